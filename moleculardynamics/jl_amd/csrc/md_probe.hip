// md_probe.hip -- libmdprobe.so: measurement aids for bench.py.  NOT part of the product ABI (nothing in
// moleculardynamics/jl_amd loads it).
//
// md_probe_fp64_rate: the fp64 vector rate this GPU sustains on independent v_fma_f64 chains, in wave64
// fp64-instruction slots per second per chip (one slot = one v_fma_f64 / v_mul_f64 / v_add_f64 issued for 64 lanes).
// bench.py prices the pair loop's ISA-counted instruction budget against it (the "valu_roofline" object): the
// datasheet figure (78.6 TFLOP/s fp64 vector = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz) assumes a clock the
// chip does not hold under this load.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) k_fma_chains(double *out, int iters, double a, double b)
{
    double x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = 1.0 + 1e-3 * (double)threadIdx.x + (double)i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += x[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int md_probe_fp64_rate(int device_id, double *slots_per_s, double *gflops)
{
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return 1;
    const int blocks = prop.multiProcessorCount * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
    const int iters = 4096;
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)blocks * 256 * sizeof(double)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) { // the first launches also bring the clock up
        (void)hipEventRecord(e0, 0);
        k_fma_chains<<<blocks, 256, 0, 0>>>(d, iters, 0.999999, 1e-6);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (hipGetLastError() != hipSuccess) return 1;
    double wave_instr = (double)blocks * 4.0 * (double)iters * 12.0;
    double sec = (double)best * 1e-3;
    if (slots_per_s) *slots_per_s = wave_instr / sec;
    if (gflops) *gflops = wave_instr * 64.0 * 2.0 / sec * 1e-9;
    return 0;
}
