"""Pin the CPU oracle against the reference's own golden outputs (CPU only).

`reference-output` fixtures: numbers the reference wrote into its checked-in run
outputs (sample_configs_gpu/*/noncuda_control/*.energy.dat:2).  `survey`
fixtures: numbers the survey stage recorded from the reference binary
(SURVEY.md 8c / BASELINE.md 2).  Both must be reproduced to every printed digit.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FX = json.load(open(os.path.join(GOLD, "fixtures.json")))

KEY = {"energy": "energy", "coulombic": "coulombic_energy", "rd": "rd_energy", "polar": "polarization_energy"}


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", sorted(FX))
def test_oracle_reproduces_reference_digits(name):
    fx = FX[name]
    if fx["n"] > 2000 and os.environ.get("MPMC_SKIP_BIG"):
        pytest.skip("big fixture skipped")
    r = oracle.energy(load(name), fx["params"])
    dec = fx["decimals"]
    for k, want in fx["expected"].items():
        if k not in KEY:
            continue
        got = r[KEY[k]]
        # the reference prints %f / %.5f: compare after rounding to the printed digits
        assert abs(got - want) <= 0.5000001 * 10 ** (-dec), (name, k, got, want)
    if "volume" in fx["expected"]:
        assert abs(r["volume"] - fx["expected"]["volume"]) < 1e-6


def test_kvector_count():
    # 709 vectors at kmax = 7
    assert oracle.lib().orc_kvector_count(7) == 709


def test_cached_oracle_equals_full_recompute_bitwise():
    """The pair-state cache (what the reference keeps in its pair list between steps) must not change a bit."""
    from mpmc_amd import synth

    s = synth.s_pol(160)
    p = dict(synth.FLAGS_POL_PRODUCTION)
    cache = oracle.Cache(160)
    rng = np.random.default_rng(3)
    pos = s["pos"].copy()
    for _ in range(5):
        s2 = dict(s, pos=pos.copy())
        a = oracle.energy(s2, p)
        b = oracle.energy(s2, p, cache=cache)
        for k in ("energy", "rd_energy", "coulombic_energy", "es_real", "polarization_energy"):
            assert a[k] == b[k], k
        m = 5 * rng.integers(0, 32)
        pos[m:m + 5] += rng.normal(scale=0.2, size=3)
    cache.close()
