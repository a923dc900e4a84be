#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mpmc_amd/csrc/kernels_coef.h"
using namespace mpmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void tiny(double *p) { if (threadIdx.x == 0) p[0] += 1.0; }
int main() {
    const int nt = 51, nv = 64 * nt;
    const size_t ncol = 192 * (size_t)nt;
    double *Srow, *Zcol, *alpha, *mu, *es, *efind, *out, *rrms, *epart;
    int *flags; unsigned long long *errmax;
    CK(hipMalloc(&Srow, 2 * ncol * nt * 8)); Zcol = Srow + ncol * nt;
    CK(hipMemset(Srow, 0, 2 * ncol * nt * 8));
    CK(hipMalloc(&alpha, nv * 8)); CK(hipMalloc(&mu, 3 * nv * 8)); CK(hipMalloc(&es, 3 * nv * 8)); CK(hipMalloc(&efind, 3 * nv * 8));
    CK(hipMalloc(&out, 3 * nv * 8)); CK(hipMalloc(&rrms, nv * 8)); CK(hipMalloc(&epart, 2 * nt * 8)); CK(hipMalloc(&flags, nv * 4)); CK(hipMalloc(&errmax, 256 * 8));
    std::vector<double> ha(nv, 1.0); std::vector<int> hf(nv, kValid);
    CK(hipMemcpy(alpha, ha.data(), nv * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(flags, hf.data(), nv * 4, hipMemcpyHostToDevice));
    CK(hipMemset(mu, 0, 3 * nv * 8)); CK(hipMemset(es, 0, 3 * nv * 8)); CK(hipMemset(errmax, 0, 256 * 8));
    CoefFinish f; f.alpha = alpha; f.flags = flags; f.mu_in = mu; f.es = es; f.ef_induced = efind; f.out = out; f.rrms = rrms; f.errmax = errmax; f.mu_final = mu; f.energy_part = epart;
    f.sp.w_new = 1.0; f.sp.w_old = 0.0; f.sp.want_rrms = 0; f.sp.err_slot = 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 200; float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(pair_finish_kernel<kSweepJacobi>, dim3(nt), dim3(64 * kCoefFinishGroups), 0, 0, nt, Srow, Zcol, f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("finish x%d back-to-back: %.2f us each\n", N, ms * 1e3 / N);
    }
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) tiny<<<1, 64>>>(Srow);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("tiny x%d: %.2f us each\n", N, ms * 1e3 / N);
    }
    return 0;
}
