// md_rccl.hpp -- RCCL bound at run time.  The library does not link librccl: the caller names the shared
// object (from Python: the copy PyTorch itself has loaded, so that one RCCL lives in the process; from a
// Julia host: /opt/rocm/lib/librccl.so) and the few entry points the step loop needs are looked up with dlsym.
// Types and enum values come from <rccl/rccl.h>.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdexcept>
#include <string>

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;

    void load(const char *path)
    {
        if (handle) return;
        const char *p = (path && path[0]) ? path : "librccl.so";
        handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!handle) throw std::runtime_error(std::string("cannot load RCCL from '") + p + "': " + dlerror());
        auto sym = [&](const char *name) {
            void *f = dlsym(handle, name);
            if (!f) throw std::runtime_error(std::string("RCCL symbol missing: ") + name);
            return f;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(handle, "ncclCommAbort"); // optional
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
    void check(ncclResult_t r, const char *what) const
    {
        if (r != ncclSuccess)
            throw std::runtime_error(std::string("RCCL ") + what + " failed: " + (GetErrorString ? GetErrorString(r) : "?"));
    }
};
