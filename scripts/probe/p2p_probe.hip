// p2p_probe.hip -- can two PROCESSES exchange data with one-sided stores and flags, no collective library?
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/p2p_probe scripts/probe/p2p_probe.hip && /tmp/p2p_probe [nproc] [iters] [bytes]
//
// The parent forks nproc children BEFORE anything touches HIP.  Every child allocates one fine-grained buffer
// (payload + a flag word), publishes its IPC handle in /tmp, opens its right-hand neighbour's, then runs
//   k_put(seq):  all blocks store a seq-dependent pattern into the NEIGHBOUR's payload (plane seq & 1: the neighbour may
//                still be reading the other one); the last block to finish
//                stores flag = seq there (release, system scope)
//   k_wait(seq): block 0 spins (bounded by a wall-clock timeout) until its OWN flag >= seq (acquire, system scope),
//                then every block checks the payload pattern
// back to back on one stream, no host wait per iteration.  Children may share one GPU (the one-GPU box: device 0 for
// all) or take device rank % ndev.  Prints per-iteration time, mismatches and timeouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/wait.h>
#include <unistd.h>

#define CHK(x)                                                                                          \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "rank %d: %s failed: %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); \
            _exit(3);                                                                                   \
        }                                                                                               \
    } while (0)
static int g_rank = -1;

struct Mail {
    unsigned long long flag;     // written by the left neighbour: last sequence number whose payload is complete
    unsigned long long pad[15];
};

__global__ void k_put(double *peer_payload, Mail *peer_mail, unsigned *done, size_t n, unsigned long long seq, int rank)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) peer_payload[i] = (double)(seq * 1000003ull + i % 977 + rank);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned prev = atomicAdd(done, 1u);
        if (prev == gridDim.x - 1) {
            *done = 0;
            __hip_atomic_store(&peer_mail->flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void k_wait(const double *payload, Mail *mail, size_t n, unsigned long long seq, int from_rank, unsigned long long *stats,
                       long long timeout_ticks)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        long long t0 = wall_clock64();
        int good = 0;
        // (after one timeout nothing waits again: the run ends within seconds whatever went wrong)
        for (; __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;) {
            unsigned long long f = __hip_atomic_load(&mail->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (f >= seq) {
                good = 1;
                break;
            }
            if (wall_clock64() - t0 > timeout_ticks) break;
            __builtin_amdgcn_s_sleep(8);
        }
        ok = good;
        if (!good && blockIdx.x == 0) atomicAdd(&stats[1], 1ull);
    }
    __syncthreads();
    if (!ok) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE); // (the loads below must not be served from a stale line)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned bad = 0;
    for (; i < n; i += stride) {
        double want = (double)(seq * 1000003ull + i % 977 + from_rank);
        double got = __builtin_nontemporal_load(&payload[i]);
        if (got != want) ++bad;
    }
    if (bad) atomicAdd(&stats[0], (unsigned long long)bad);
}

static void child(int rank, int nproc, int iters, size_t bytes, const char *tag)
{
    g_rank = rank;
    int ndev = 0;
    CHK(hipGetDeviceCount(&ndev));
    int dev = rank % ndev;
    CHK(hipSetDevice(dev));
    size_t n = bytes / 8;
    char *buf = nullptr;
    size_t total = sizeof(Mail) + 2 * n * 8; // two payload planes, by the parity of the sequence number
    CHK(hipExtMallocWithFlags((void **)&buf, total, hipDeviceMallocFinegrained));
    CHK(hipMemset(buf, 0, total));
    CHK(hipDeviceSynchronize());
    hipIpcMemHandle_t mine;
    CHK(hipIpcGetMemHandle(&mine, buf));
    std::string base = std::string("/tmp/p2p_probe_") + tag + "_";
    {
        std::string tmp = base + std::to_string(rank) + ".tmp", fin = base + std::to_string(rank);
        FILE *f = fopen(tmp.c_str(), "wb");
        fwrite(&mine, sizeof mine, 1, f);
        fclose(f);
        rename(tmp.c_str(), fin.c_str());
    }
    int right = (rank + 1) % nproc, left = (rank + nproc - 1) % nproc;
    char *peer = buf;
    if (nproc > 1) {
        hipIpcMemHandle_t theirs;
        std::string fin = base + std::to_string(right);
        for (int tries = 0;; ++tries) {
            FILE *f = fopen(fin.c_str(), "rb");
            if (f) {
                size_t got = fread(&theirs, sizeof theirs, 1, f);
                fclose(f);
                if (got == 1) break;
            }
            if (tries > 3000) {
                fprintf(stderr, "rank %d: neighbour's handle never appeared\n", rank);
                _exit(4);
            }
            usleep(10000);
        }
        CHK(hipIpcOpenMemHandle((void **)&peer, theirs, hipIpcMemLazyEnablePeerAccess));
    }
    unsigned *done;
    unsigned long long *stats;
    CHK(hipMalloc(&done, 4));
    CHK(hipMemset(done, 0, 4));
    CHK(hipMalloc(&stats, 16));
    CHK(hipMemset(stats, 0, 16));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const long long timeout = 100000000ll * 3; // wall_clock64: 100 MHz -> 3 s
    int blocks = (int)((n + 255) / 256);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    double *peer_payload = (double *)(peer + sizeof(Mail));
    Mail *peer_mail = (Mail *)peer;
    const double *my_payload = (const double *)(buf + sizeof(Mail));
    Mail *my_mail = (Mail *)buf;
    for (int phase = 0; phase < 2; ++phase) {
        int it0 = phase == 0 ? 1 : 1 + 20, it1 = phase == 0 ? 20 : 20 + iters;
        if (phase == 1) CHK(hipEventRecord(e0, st));
        for (int it = it0; it <= it1; ++it) {
            // (a real step kernel would sit here)
            hipLaunchKernelGGL(k_put, dim3(blocks), dim3(256), 0, st, peer_payload + (size_t)(it & 1) * n, peer_mail, done, n, (unsigned long long)it, rank);
            hipLaunchKernelGGL(k_wait, dim3(blocks), dim3(256), 0, st, my_payload + (size_t)(it & 1) * n, my_mail, n, (unsigned long long)it, left, stats,
                               timeout);
        }
        if (phase == 1) CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
    }
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CHK(hipMemcpy(h, stats, 16, hipMemcpyDeviceToHost));
    printf("rank %d/%d dev %d: %d iterations of put+wait, %zu B payload: %.2f us/iteration, mismatches %llu, timeouts %llu\n", rank, nproc,
           dev, iters, n * 8, 1e3 * ms / iters, h[0], h[1]);
    fflush(stdout);
    if (nproc > 1) CHK(hipIpcCloseMemHandle(peer));
    _exit((h[0] || h[1]) ? 5 : 0);
}

int main(int argc, char **argv)
{
    int nproc = argc > 1 ? atoi(argv[1]) : 2;
    int iters = argc > 2 ? atoi(argv[2]) : 200;
    size_t bytes = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)1400000;
    char tag[32];
    snprintf(tag, sizeof tag, "%d", (int)getpid());
    pid_t pids[8];
    if (nproc < 1 || nproc > 6) return 2;
    for (int r = 0; r < nproc; ++r) {
        pids[r] = fork();
        if (pids[r] == 0) child(r, nproc, iters, bytes, tag);
    }
    int rc = 0;
    for (int r = 0; r < nproc; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    for (int r = 0; r < nproc; ++r) {
        std::string f = std::string("/tmp/p2p_probe_") + tag + "_" + std::to_string(r);
        unlink(f.c_str());
    }
    printf("probe %s\n", rc ? "FAILED" : "ok");
    return rc;
}
