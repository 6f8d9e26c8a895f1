// Gradient-exchange entry of the C-ABI (last row): one bucket of the flat gradient arena, summed in place over
// the caller's RCCL communicator on the caller's stream.  Replaces (reference): the all-reduce torch DDP issues per bucket
// under pl.trainer.strategy=ddp (config/pl/default.yaml:2, README.md:84-94).
//
// The library does not link librccl: a process that never exchanges gradients (sampling, single-GPU training) must not pay
// for -- or fail on -- a communication library, and a PyTorch process already holds ITS OWN copy (torch/lib/librccl.so) whose
// communicators are only valid inside that copy.  The entry points are therefore resolved at first use: an instance the
// host named with sgd_exchange_bind(), else one that is already loaded in the process (RTLD_NOLOAD), else the system's.
// Host code only; no kernels here.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>

#include <hip/hip_runtime.h>

#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {
// rccl.h: ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
// ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0 (the values are ABI of NCCL 2.x / RCCL and have never changed)
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*errstr_fn)(int);
std::mutex g_mu;
void* g_handle = nullptr;
allreduce_fn g_allreduce = nullptr;
errstr_fn g_errstr = nullptr;

int bind_locked(const char* path) {
    void* h = nullptr;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);     // the process's own copy first
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        fprintf(stderr, "sgdm_hip: sgd_allreduce_bucket: no RCCL library (%s): %s\n", path && *path ? path : "librccl.so.1", dlerror());
        return SGD_ERR_LAUNCH;
    }
    allreduce_fn f = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
    if (!f) {
        fprintf(stderr, "sgdm_hip: sgd_allreduce_bucket: the RCCL library does not export ncclAllReduce\n");
        dlclose(h);
        return SGD_ERR_LAUNCH;
    }
    g_handle = h;
    g_allreduce = f;
    g_errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
    return SGD_OK;
}
}  // namespace

extern "C" int sgd_exchange_bind(const char* librccl_path) {
    std::lock_guard<std::mutex> lk(g_mu);
    return bind_locked(librccl_path);
}

extern "C" int sgd_allreduce_bucket(void* nccl_comm, float* bucket, int64_t count, void* stream) {
    if (!nccl_comm || !bucket || count <= 0) return SGD_ERR_ARG;
    allreduce_fn f;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_allreduce) {
            const int rc = bind_locked(nullptr);
            if (rc != SGD_OK) return rc;
        }
        f = g_allreduce;
    }
    const int rc = f(bucket, bucket, (size_t)count, /* ncclFloat32 */ 7, /* ncclSum */ 0, nccl_comm, (hipStream_t)stream);
    if (rc != 0) {
        fprintf(stderr, "sgdm_hip: ncclAllReduce failed: %s (%d)\n", g_errstr ? g_errstr(rc) : "?", rc);
        return SGD_ERR_LAUNCH;
    }
    return SGD_OK;
}
