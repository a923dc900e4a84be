// handoff_probe.hip -- latency of a data-is-the-flag hand-off chain between workgroups (one 16-byte granule per hop):
//   placement: "rr" = consecutive links on consecutive workgroups (dealt round-robin over the 8 XCDs), "xcd" = all links on
//              ONE XCD (roles by HW_REG_XCC_ID); store: sc1 (write-through) or plain; load: sc1 (agent scope).
// Build: hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
static const unsigned long long kSent = 0x7ff8dead7ff8deadull;
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int PLAIN, int XCD>
__global__ __launch_bounds__(64) void chain(unsigned long long *slots, unsigned *ticket, int n, unsigned long long *stamps, int pad_lds) {
    extern __shared__ double lds[];
    if (pad_lds && threadIdx.x == 0) lds[0] = 0.0;
    __shared__ int s_k;
    if (threadIdx.x == 0) {
        int k = -1;
        if (!XCD || xcc_id() == 0) k = (int)atomicAdd(ticket, 1u);
        s_k = k;
    }
    __syncthreads();
    const int k = s_k;
    if (k < 0 || k >= n) return;
    unsigned long long v = 1;
    if (k > 0) {
        const unsigned long long *q = slots + 2 * (size_t)(k - 1);
        for (unsigned it = 0; it < (1u << 24); ++it) {
            v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != kSent) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (threadIdx.x == 0) {
        unsigned long long *dst = slots + 2 * (size_t)k;
        if (PLAIN) {
            dst[0] = v + 1;
            __builtin_amdgcn_s_waitcnt(0);
        } else
            __hip_atomic_store(dst, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamps[k] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---- the chain kernel's hand-off shape: a block of 192 doubles published by 96 lanes with 16-byte sc1 stores, polled by
// the 8 waves of the next workgroup (MODE 0: 24 lanes per wave, three loads in flight; 1: the same, one load at a time;
// 2: ONE wave polls all 192 doubles -- 3 loads per lane -- and the others wait at a barrier), with a barrier between the
// poll and the publication as in the kernel.  EXTRA: every block is also polled by the workgroups 2 and 3 hops behind
// (the auxiliary workgroups of the chain), which publish nothing.
__device__ __forceinline__ void st16(double *p, double a, double b) {
    typedef double __attribute__((ext_vector_type(2))) d2_t;
    const d2_t v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long ldq(const double *p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ double poll(const double *p) {
    if (MODE == 0) {
        unsigned long long v0 = ldq(p);
        __builtin_amdgcn_s_sleep(1);
        unsigned long long v1 = ldq(p);
        __builtin_amdgcn_s_sleep(1);
        unsigned long long v2 = ldq(p);
        for (unsigned it = 0; it < (1u << 24); ++it) {
            if (v0 != kSent) return __longlong_as_double((long long)v0);
            v0 = v1;
            v1 = v2;
            v2 = ldq(p);
            __builtin_amdgcn_s_sleep(1);
        }
        return 0.0;
    }
    for (unsigned it = 0; it < (1u << 24); ++it) {
        const unsigned long long v = ldq(p);
        if (v != kSent) return __longlong_as_double((long long)v);
    }
    return 0.0;
}
template <int MODE, int EXTRA>
__global__ __launch_bounds__(512) void wide(double *blocks, unsigned *ticket, int n, unsigned long long *stamps) {
    __shared__ int s_k;
    __shared__ double buf[8 * 192];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_k = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    int k = s_k, role = 0;
    if (EXTRA) {  // tickets: main(k), then its two followers
        role = k % 3;
        k = k / 3;
    }
    if (k >= n) return;
    const int src = role == 0 ? k - 1 : k - 1 - role;  // followers poll older blocks (and publish nothing)
    double v = 1.0;
    if (src >= 0) {
        const double *q = blocks + 192 * (size_t)src;
        if (MODE == 2) {
            if (w == 0) {
                const double a = poll<1>(q + lane), b = poll<1>(q + 64 + lane), c = poll<1>(q + 128 + lane);
                buf[lane] = a;
                buf[64 + lane] = b;
                buf[128 + lane] = c;
            }
            __syncthreads();
            v = buf[(tid * 7) % 192];
        } else {
            if (lane < 24) buf[192 * w + lane] = poll<MODE>(q + 64 * (lane % 3) + 8 * w + lane / 3);
            v = buf[192 * w + (lane % 24)];
        }
    }
    buf[192 * w + 32 + (lane & 31)] = v;  // (some LDS traffic and the barrier the real kernel has here)
    __syncthreads();
    if (role != 0) return;
    if (tid < 96) st16(blocks + 192 * (size_t)k + 2 * tid, v + 1.0, v + 1.0);
    if (tid == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();
}

// ---- which side of the wide hand-off costs the time?  PUB 0: 96 lanes x 16 B (waves 0, 1), 1: every wave stores the 24
// doubles [24 w, 24 w + 24) with 12 lanes x 16 B, 2: ONE line only (4 lanes x 16 B); CON 1: planar addresses (3 lines 512 B
// apart per wave), one load at a time, 3: wave-major [24 w + lane] (3 adjacent lines), one at a time, 5: wave-major, three in
// flight, 4: every wave polls the first line only (8 lanes), 6: wave-major, 12 lanes x 16-byte loads
template <int PUB, int CON>
__global__ __launch_bounds__(512) void shape(double *blocks, unsigned *ticket, int n, unsigned long long *stamps) {
    __shared__ int s_k;
    __shared__ double buf[8 * 192];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_k = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int k = s_k;
    if (k >= n) return;
    double v = 1.0;
    if (k > 0) {
        const double *q = blocks + 192 * (size_t)(k - 1);
        if (CON == 1) {
            if (lane < 24) buf[192 * w + lane] = poll<1>(q + 64 * (lane % 3) + 8 * w + lane / 3);
        } else if (CON == 3) {
            if (lane < 24) buf[192 * w + lane] = poll<1>(q + 24 * w + lane);
        } else if (CON == 5) {
            if (lane < 24) buf[192 * w + lane] = poll<0>(q + 24 * w + lane);
        } else if (CON == 4) {
            if (lane < 8) buf[192 * w + lane] = poll<1>(q + lane);
        } else if (CON == 6) {
            if (lane < 12) {
                typedef unsigned long long __attribute__((ext_vector_type(2))) u2_t;
                const double *a = q + 24 * w + 2 * lane;
                u2_t r;
                for (unsigned it = 0; it < (1u << 24); ++it) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(a) : "memory");
                    if (r.x != kSent && r.y != kSent) break;
                }
                buf[192 * w + 2 * lane] = __longlong_as_double((long long)r.x);
                buf[192 * w + 2 * lane + 1] = __longlong_as_double((long long)r.y);
            }
        }
        if (CON == 7 || CON == 8 || CON == 9) {  // ONE wave polls (7: 48 lanes x two 16-byte loads in flight together; 8: 64 lanes x
                                                 // three 8-byte loads together; 9: the first line only), the others wait at a barrier
            if (w == 0) {
                typedef unsigned long long __attribute__((ext_vector_type(2))) u2_t;
                if (CON == 7 && lane < 48) {
                    const double *a = q + 4 * lane;
                    u2_t r0, r1;
                    for (unsigned it = 0; it < (1u << 24); ++it) {
                        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(r0), "=&v"(r1) : "v"(a) : "memory");
                        if (r0.x != kSent && r0.y != kSent && r1.x != kSent && r1.y != kSent) break;
                    }
                    buf[4 * lane] = __longlong_as_double((long long)r0.x);
                    buf[4 * lane + 1] = __longlong_as_double((long long)r0.y);
                    buf[4 * lane + 2] = __longlong_as_double((long long)r1.x);
                    buf[4 * lane + 3] = __longlong_as_double((long long)r1.y);
                } else if (CON == 8) {
                    unsigned long long a, b, c;
                    for (unsigned it = 0; it < (1u << 24); ++it) {
                        a = ldq(q + lane);
                        b = ldq(q + 64 + lane);
                        c = ldq(q + 128 + lane);
                        if (a != kSent && b != kSent && c != kSent) break;
                    }
                    buf[lane] = __longlong_as_double((long long)a);
                    buf[64 + lane] = __longlong_as_double((long long)b);
                    buf[128 + lane] = __longlong_as_double((long long)c);
                } else if (CON == 9 && lane < 8) {
                    buf[lane] = poll<1>(q + lane);
                }
            }
            __syncthreads();
            v = buf[(lane % 8)];
        } else
            v = buf[192 * w + (lane % 8)];
    }
    buf[192 * w + 32 + (lane & 31)] = v;
    __syncthreads();
    if (PUB == 0) {
        if (tid < 96) st16(blocks + 192 * (size_t)k + 2 * tid, v + 1.0, v + 1.0);
    } else if (PUB == 1) {
        if (lane < 12) st16(blocks + 192 * (size_t)k + 24 * w + 2 * lane, v + 1.0, v + 1.0);
    } else {
        if (tid < 4) st16(blocks + 192 * (size_t)k + 2 * tid, v + 1.0, v + 1.0);
    }
    if (tid == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();
}
__global__ void arm_wide(double *blocks, unsigned *ticket, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 192 * n) reinterpret_cast<unsigned long long *>(blocks)[i] = kSent;
    if (i == 0) *ticket = 0;
}

__global__ void arm(unsigned long long *slots, unsigned *ticket, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n) slots[i] = kSent;
    if (i == 0) *ticket = 0;
}
int main() {
    const int n = 32;
    unsigned long long *slots, *stamps;
    unsigned *ticket;
    hipMalloc(&slots, 2 * n * 8);
    hipMalloc(&stamps, n * 8);
    hipMalloc(&ticket, 4);
    std::vector<unsigned long long> h(n);
    auto run = [&](const char *name, auto kern, int grid, int lds) {
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(arm, dim3(1), dim3(256), 0, 0, slots, ticket, n);
            hipMemset(stamps, 0, n * 8);
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, slots, ticket, n, stamps, lds > 0);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), stamps, n * 8, hipMemcpyDeviceToHost);
            bool ok = true;
            for (int k = 0; k < n; ++k) ok &= (h[k] != 0);
            if (!ok) { printf("%s: chain incomplete\n", name); return; }
            const double per = (double)(h[n - 1] - h[4]) * 0.01 / (n - 1 - 4);  // 100 MHz ticks -> us per hop, past the ramp
            if (rep >= 2) { best = per < best ? per : best; sum += per; }
        }
        printf("%-46s %.3f us per hop (min), %.3f mean\n", name, best, sum / 18);
    };
    run("round-robin placement, sc1 store", chain<0, 0>, n, 0);
    run("round-robin placement, plain store", chain<1, 0>, n, 0);
    run("one XCD, sc1 store", chain<0, 1>, 8 * n, 0);
    run("one XCD, plain store", chain<1, 1>, 8 * n, 0);
    run("one XCD, plain store, 1 workgroup per CU (LDS)", chain<1, 1>, 8 * n, 150 * 1024);
    run("one XCD, sc1 store, 1 workgroup per CU (LDS)", chain<0, 1>, 8 * n, 150 * 1024);
    run("round-robin, sc1 store, 1 workgroup per CU", chain<0, 0>, n, 150 * 1024);
    double *blocks;
    hipMalloc(&blocks, 192 * n * 8);
    auto runw = [&](const char *name, auto kern, int grid) {
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(arm_wide, dim3((192 * n + 255) / 256), dim3(256), 0, 0, blocks, ticket, n);
            hipMemset(stamps, 0, n * 8);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, blocks, ticket, n, stamps);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), stamps, n * 8, hipMemcpyDeviceToHost);
            bool ok = true;
            for (int k = 0; k < n; ++k) ok &= (h[k] != 0);
            if (!ok) { printf("%s: chain incomplete\n", name); return; }
            const double per = (double)(h[n - 1] - h[4]) * 0.01 / (n - 1 - 4);
            if (rep >= 2) { best = per < best ? per : best; sum += per; }
        }
        printf("%-62s %.3f us per hop (min), %.3f mean\n", name, best, sum / 18);
    };
    runw("shape: 96x16B store, planar polls (1 at a time)", shape<0, 1>, n);
    runw("shape: 96x16B store, every wave polls ONE line", shape<0, 4>, n);
    runw("shape: ONE line stored, every wave polls it", shape<2, 4>, n);
    runw("shape: 96x16B store, wave-major polls", shape<0, 3>, n);
    runw("shape: 8 waves x 12 x 16B store, wave-major polls", shape<1, 3>, n);
    runw("shape: 8 waves x 12 x 16B store, wave-major polls, 3 in flight", shape<1, 5>, n);
    runw("shape: 8 waves x 12 x 16B store, wave-major 16-byte polls", shape<1, 6>, n);
    runw("shape: 8 waves x 12 x 16B store, every wave polls ONE line", shape<1, 4>, n);
    runw("shape: 96x16B store, ONE wave polls: 48 lanes x 2 x 16 B", shape<0, 7>, n);
    runw("shape: 96x16B store, ONE wave polls: 64 lanes x 3 x 8 B", shape<0, 8>, n);
    runw("shape: 96x16B store, ONE wave polls the first line", shape<0, 9>, n);
    runw("shape: ONE line stored, ONE wave polls it", shape<2, 9>, n);
    runw("192 doubles, 8 waves x 24 lanes, 3 loads in flight", wide<0, 0>, n);
    runw("192 doubles, 8 waves x 24 lanes, 1 load at a time", wide<1, 0>, n);
    runw("192 doubles, one wave polls all, barrier", wide<2, 0>, n);
    runw("192 doubles, 8 x 24, 3 in flight, + 2 follower workgroups/block", wide<0, 1>, 3 * n);
    runw("192 doubles, 8 x 24, 1 at a time, + 2 follower workgroups/block", wide<1, 1>, 3 * n);
    runw("192 doubles, one wave polls, + 2 follower workgroups/block", wide<2, 1>, 3 * n);
    return 0;
}
