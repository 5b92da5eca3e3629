// m324_comm_*: the collectives of the path behind the C ABI, on RCCL (xGMI inside a node).
//
// Two exchanges exist on this path: the all-reduce of the flat fp32 gradient buffer in the data-
// parallel training step (the reference's DDP reducer, train.py:88-89,159-166) and the all-gather of the k|v projection
// of every global block / of the final [T, N, 3] offsets in frame- and clip-parallel inference.  The Python host of this
// repo issues them through torch.distributed (backend "nccl" = RCCL), which owns the process group the reference's
// setup.py:134-140 creates; a host that binds libm324 WITHOUT torch gets the same two collectives here.
//
// libm324.so neither links RCCL nor needs its headers: the library is resolved at m324_comm_init time, in this order:
// (1) the path in M324_RCCL_LIB, (2) a copy ALREADY LOADED into the process -- global symbols first, then
// dlopen(RTLD_NOLOAD), which also finds the librccl.so a torch process loaded with RTLD_LOCAL (two RCCL instances must
// not be mixed) --, (3) librccl.so from the loader path.  The handful of RCCL types and enum values used here are
// restated below (they are RCCL's ABI: nccl.h of RCCL 2.x), so the kernels build and load on a box without RCCL.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
}

struct m324_comm {
    ncclComm_t comm;
    int rank, world;
};

namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    bool ok;
};

const Rccl* rccl() {
    static const Rccl r = [] {
        Rccl x{};
        void* h = nullptr;
        if (const char* forced = getenv("M324_RCCL_LIB")) h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        if (!h && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = RTLD_DEFAULT;
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);         // already mapped (e.g. by torch, RTLD_LOCAL): reuse THAT copy
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(h, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(h, "ncclCommDestroy");
        x.AllReduce = (decltype(x.AllReduce))dlsym(h, "ncclAllReduce");
        x.AllGather = (decltype(x.AllGather))dlsym(h, "ncclAllGather");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(h, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.GetErrorString;
        return x;
    }();
    return &r;
}

#define M324_RCCL(call, what)                                                                       \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) M324_FAIL(M324_ERR_HIP, "%s: %s", what, rccl()->GetErrorString(r_)); \
    } while (0)

bool nccl_type(int dtype, ncclDataType_t* t) {
    if (dtype == M324_F32) { *t = ncclFloat32; return true; }
    if (dtype == M324_BF16) { *t = ncclBfloat16; return true; }
    return false;
}
}  // namespace

extern "C" int m324_comm_unique_id(char* hex, int n) {
    M324_REQUIRE(hex && n >= 2 * NCCL_UNIQUE_ID_BYTES + 1, "m324_comm_unique_id: buffer must hold %d characters", 2 * NCCL_UNIQUE_ID_BYTES + 1);
    M324_REQUIRE(rccl()->ok, "m324_comm: RCCL (librccl.so) is not available in this process");
    ncclUniqueId id;
    M324_RCCL(rccl()->GetUniqueId(&id), "ncclGetUniqueId");
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    return M324_OK;
}

extern "C" int m324_comm_init(m324_comm** out, const char* unique_id_hex, int rank, int world) {
    M324_REQUIRE(out && unique_id_hex && world >= 1 && rank >= 0 && rank < world, "m324_comm_init: bad arguments");
    M324_REQUIRE(strlen(unique_id_hex) == 2 * NCCL_UNIQUE_ID_BYTES, "m324_comm_init: the id must be %d hex characters", 2 * NCCL_UNIQUE_ID_BYTES);
    M324_REQUIRE(rccl()->ok, "m324_comm: RCCL (librccl.so) is not available in this process");
    ncclUniqueId id;
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) {
        unsigned v = 0;
        M324_REQUIRE(sscanf(unique_id_hex + 2 * i, "%2x", &v) == 1, "m324_comm_init: malformed id");
        id.internal[i] = (char)v;
    }
    ncclComm_t c;
    M324_RCCL(rccl()->CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    *out = new m324_comm{c, rank, world};
    return M324_OK;
}

extern "C" int m324_comm_allreduce(m324_comm* c, void* buf, long count, int dtype, int average, void* stream) {
    ncclDataType_t t;
    M324_REQUIRE(c && buf && count > 0 && nccl_type(dtype, &t), "m324_comm_allreduce: bad arguments");
    M324_RCCL(rccl()->AllReduce(buf, buf, (size_t)count, t, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
    return M324_OK;
}

extern "C" int m324_comm_allgather(m324_comm* c, const void* send, void* recv, long count_per_rank, int dtype, void* stream) {
    ncclDataType_t t;
    M324_REQUIRE(c && send && recv && count_per_rank > 0 && nccl_type(dtype, &t), "m324_comm_allgather: bad arguments");
    M324_RCCL(rccl()->AllGather(send, recv, (size_t)count_per_rank, t, c->comm, (hipStream_t)stream), "ncclAllGather");
    return M324_OK;
}

extern "C" int m324_comm_destroy(m324_comm* c) {
    if (!c) return M324_OK;
    ncclResult_t r = rccl()->ok ? rccl()->CommDestroy(c->comm) : ncclSuccess;
    delete c;
    if (r != ncclSuccess) M324_FAIL(M324_ERR_HIP, "ncclCommDestroy: %s", rccl()->GetErrorString(r));
    return M324_OK;
}
