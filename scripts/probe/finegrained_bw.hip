// finegrained_bw.hip -- store / load bandwidth of one kernel into device memory of the three allocation kinds
// (hipMalloc, hipExtMallocWithFlags fine-grained, uncached), local device.  Sizes the direct peer exchange's cost model.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/fg scripts/probe/finegrained_bw.hip && /tmp/fg
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_store(double *p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = v + (double)i;
}
__global__ void k_store_nt(double *p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) __builtin_nontemporal_store(v + (double)i, &p[i]);
}
__global__ void k_store2(double2 *p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = make_double2(v + (double)i, v);
}
__global__ void k_load(const double *p, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    double a = 0;
    for (; i < n; i += s) a += p[i];
    if (a == 1.2345) out[0] = a;
}
int main()
{
    const size_t n = 3000000 / 8 * 8 / 8; // ~3 MB: two faces' records
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    double *out;
    CHK(hipMalloc(&out, 8));
    const char *names[3] = {"hipMalloc (coarse)", "fine-grained", "uncached"};
    for (int kind = 0; kind < 3; ++kind) {
        double *p = nullptr;
        if (kind == 0) CHK(hipMalloc(&p, n * 8));
        if (kind == 1) CHK(hipExtMallocWithFlags((void **)&p, n * 8, hipDeviceMallocFinegrained));
        if (kind == 2) CHK(hipExtMallocWithFlags((void **)&p, n * 8, hipDeviceMallocUncached));
        for (int blocks : {64, 240, 1024, 4096}) {
            float ms[4] = {0, 0, 0, 0};
            for (int var = 0; var < 4; ++var) {
                for (int rep = 0; rep < 12; ++rep) {
                    if (rep == 2) CHK(hipEventRecord(e0));
                    if (var == 0) k_store<<<blocks, 256>>>(p, n, 1.0);
                    if (var == 1) k_store_nt<<<blocks, 256>>>(p, n, 1.0);
                    if (var == 2) k_store2<<<blocks, 256>>>((double2 *)p, n / 2, 1.0);
                    if (var == 3) k_load<<<blocks, 256>>>(p, n, out);
                }
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms[var], e0, e1));
                ms[var] /= 10;
            }
            printf("%-20s %5d blocks: store %6.1f us (%5.0f GB/s)  nt-store %6.1f us  16B-store %6.1f us  load %6.1f us (%5.0f GB/s)\n", names[kind],
                   blocks, 1e3 * ms[0], n * 8 / (ms[0] * 1e6), 1e3 * ms[1], 1e3 * ms[2], 1e3 * ms[3], n * 8 / (ms[3] * 1e6));
        }
        CHK(hipFree(p));
    }
    return 0;
}
