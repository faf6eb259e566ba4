// nccl_shim.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that carries the native window loop's collectives
// (md_dom_run_window: ncclAllReduce of the violation flag and of K/U/W, grouped ncclSend/ncclRecv of the halo
// coordinates) between several PROCESSES THAT SHARE ONE GPU, through POSIX shared memory.  RCCL itself needs one GPU
// per rank, which the one-GPU test box cannot give; with this shim the library's own control flow -- issue order with
// two ranks (left and right neighbour are the same peer), the global violation flag, window skipping, prune steps --
// runs with 2 and 3 real ranks.  libmdhip binds it exactly like RCCL: dlopen(path) + dlsym of the same symbols
// (csrc/md_rccl.hpp).  Every call is synchronous: hipStreamSynchronize, host copy, exchange, copy back.
// ncclCommAbort on any rank sets a shared flag that every blocked (and every later) call of every rank returns an error
// on, as RCCL's does: the library's fail-fast path (DomAbortGuard) can be tested with real peers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr size_t CAP = 4u << 20; // bytes per message slot
constexpr int SLOTS = 4;
constexpr int MAXR = 8;

struct Box { // single-producer single-consumer ring, src -> dst
    std::atomic<uint64_t> head, tail;
    size_t bytes[SLOTS];
    char data[SLOTS][CAP];
};
struct Shared {
    std::atomic<int> bar_count, bar_gen;
    std::atomic<int> aborted; // set by ncclCommAbort on ANY rank: every blocked or later call of every rank fails
                              // (RCCL's behaviour after an abort: peers error out instead of hanging)
    double ar[MAXR][64];
    Box box[MAXR][MAXR];
};
struct Comm {
    int rank, nranks;
    Shared *sh;
};

bool barrier(Comm *c) // false: the communicator was aborted while waiting
{
    int gen = c->sh->bar_gen.load();
    if (c->sh->bar_count.fetch_add(1) == c->nranks - 1) {
        c->sh->bar_count.store(0);
        c->sh->bar_gen.fetch_add(1);
    } else {
        while (c->sh->bar_gen.load() == gen) {
            if (c->sh->aborted.load()) return false;
            sched_yield();
        }
    }
    return !c->sh->aborted.load();
}
size_t tsize(ncclDataType_t t) { return (t == ncclFloat64 || t == ncclInt64 || t == ncclUint64) ? 8 : 4; }
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) return ncclSystemError;
    close(fd);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks > MAXR) return ncclInvalidArgument;
    char name[64] = "/mdshim_";
    for (int i = 0; i < 12; ++i) snprintf(name + 8 + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm *c = new Comm{rank, nranks, (Shared *)p};
    barrier(c);
    if (rank == 0) shm_unlink(name); // everybody has it mapped: nothing is left behind in /dev/shm
    barrier(c);
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    munmap(c->sh, sizeof(Shared));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    ((Comm *)comm)->sh->aborted.store(1); // releases every peer spinning in a barrier, Send or Recv -- with an error
    return ncclCommDestroy(comm);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (count > 64) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    size_t nb = count * tsize(dt);
    if (hipMemcpy(c->sh->ar[c->rank], sendbuff, nb, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclRemoteError;
    char out[64 * 8];
    if (dt == ncclFloat64 && op == ncclSum) {
        double *o = (double *)out;
        for (size_t i = 0; i < count; ++i) {
            double s = 0.0;
            for (int r = 0; r < c->nranks; ++r) s += c->sh->ar[r][i]; // rank order: the same bits on every rank
            o[i] = s;
        }
    } else if (dt == ncclInt32 && op == ncclMin) {
        int32_t *o = (int32_t *)out;
        for (size_t i = 0; i < count; ++i) {
            int32_t m = ((int32_t *)c->sh->ar[0])[i];
            for (int r = 1; r < c->nranks; ++r) m = ((int32_t *)c->sh->ar[r])[i] < m ? ((int32_t *)c->sh->ar[r])[i] : m;
            o[i] = m;
        }
    } else {
        return ncclInvalidArgument;
    }
    if (!barrier(c)) return ncclRemoteError;
    if (hipMemcpy(recvbuff, out, nb, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    size_t nb = count * tsize(dt);
    if (nb > CAP) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    Box &b = c->sh->box[c->rank][peer];
    while (b.head.load() - b.tail.load() >= (uint64_t)SLOTS) {
        if (c->sh->aborted.load()) return ncclRemoteError;
        sched_yield();
    }
    if (c->sh->aborted.load()) return ncclRemoteError;
    int s = (int)(b.head.load() % SLOTS);
    if (hipMemcpy(b.data[s], sendbuff, nb, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    b.bytes[s] = nb;
    b.head.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    size_t nb = count * tsize(dt);
    (void)stream;
    Box &b = c->sh->box[peer][c->rank];
    while (b.head.load() == b.tail.load()) {
        if (c->sh->aborted.load()) return ncclRemoteError;
        sched_yield();
    }
    int s = (int)(b.tail.load() % SLOTS);
    if (b.bytes[s] != nb) {
        fprintf(stderr, "[nccl_shim] rank %d: message from %d has %zu bytes, receive expects %zu\n", c->rank, peer, b.bytes[s], nb);
        return ncclInvalidUsage;
    }
    if (hipMemcpy(recvbuff, b.data[s], nb, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    b.tail.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "nccl_shim error"; }
}
