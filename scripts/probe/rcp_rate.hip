// Throughput of v_rcp_f64 vs v_fma_f64 on gfx950: 8 independent chains per lane, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.5;          // rcp + add
            if (MODE == 1) a[i] = __builtin_fma(a[i], 0.999, 1.5) + 1.5;       // fma + add
            if (MODE == 2) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]) + 1.5; // cvt + rcp_f32 + cvt + add
            if (MODE == 3) a[i] = a[i] + 1.5;                                   // add only
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double *d;
    hipMalloc(&d, 1024 * 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<1024, 256>>>(d, iters);
            if (mode == 1) k<1><<<1024, 256>>>(d, iters);
            if (mode == 2) k<2><<<1024, 256>>>(d, iters);
            if (mode == 3) k<3><<<1024, 256>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions of the timed op group per SIMD: 1024 blocks * 4 waves / 1024 SIMDs = 4 waves per SIMD
        double groups = 4.0 * iters * 8;
        printf("mode %d: %.3f ms  -> %.1f cycles per (op group) per wave at 2.4 GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / groups);
    }
    return 0;
}
