// The one collective of the path, for callers without torch.distributed (mc_comm_init /
// mc_allreduce_rank): every rank holds the partial positions of ITS shard of the samples for all T test
// passwords; the guess-number estimates are their sum over the ranks.  The reference has no counterpart
// (it is single-process); the sum replaces nothing of it but makes `positions[idx - 1]`
// (lib4mc/MonteCarloLib.py:31-32) a distributed quantity.
//
// RCCL is bound at run time (dlopen: a copy the process already holds -- PyTorch brings one -- is preferred over
// loading a second), so the library loads and everything else works on a machine without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "mc_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
bool g_ok = false;

void bind_rccl() {
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if (!h) h = dlopen(n, RTLD_NOLOAD | RTLD_NOW);
    for (const char *n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_rccl.handle = h;
    auto sym = [&](const char *s) { return dlsym(h, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy &&
           g_rccl.GetErrorString && g_rccl.GroupStart && g_rccl.GroupEnd;
}

bool have_rccl(const char *who) {
    std::call_once(g_once, bind_rccl);
    if (!g_ok) {
        const char *why = dlerror();      // (one call: reading the message clears it)
        mc_set_error("%s: RCCL (librccl.so) could not be loaded: %s", who, why ? why : "symbols missing");
    }
    return g_ok;
}

int fail(const char *who, const char *what, ncclResult_t r) {
    mc_set_error("%s: %s failed: %s", who, what, g_rccl.GetErrorString(r));
    return MC_ERR_HIP;
}

}  // namespace

struct mc_comm {
    ncclComm_t comm;
    int rank, world;
};

static_assert(MC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id the caller carries between ranks is RCCL's");

extern "C" int mc_comm_unique_id(unsigned char *id) {
    if (!id) {
        mc_set_error("mc_comm_unique_id: bad argument");
        return MC_ERR_ARG;
    }
    if (!have_rccl("mc_comm_unique_id")) return MC_ERR_HIP;
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return fail("mc_comm_unique_id", "ncclGetUniqueId", r);
    std::memcpy(id, u.internal, MC_COMM_ID_BYTES);
    return MC_OK;
}

extern "C" int mc_comm_init(const unsigned char *id, int rank, int world, mc_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) {
        mc_set_error("mc_comm_init: bad argument (id %p, rank %d of %d)", (const void *)id, rank, world);
        return MC_ERR_ARG;
    }
    if (!have_rccl("mc_comm_init")) return MC_ERR_HIP;
    ncclUniqueId u;
    std::memcpy(u.internal, id, MC_COMM_ID_BYTES);
    mc_comm *c = new (std::nothrow) mc_comm();
    if (!c) return MC_ERR_NOMEM;
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);  // on the calling thread's current device
    if (r != ncclSuccess) {
        delete c;
        return fail("mc_comm_init", "ncclCommInitRank", r);
    }
    *out = c;
    return MC_OK;
}

extern "C" int mc_allreduce_rank(mc_comm *c, double *pos, uint64_t *cnt, uint64_t t, void *stream) {
    if (!c || (t && !pos)) {
        mc_set_error("mc_allreduce_rank: bad argument");
        return MC_ERR_ARG;
    }
    if (t == 0) return MC_OK;
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return fail("mc_allreduce_rank", "ncclGroupStart", r);
    r = g_rccl.AllReduce(pos, pos, (size_t)t, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream);
    if (r == ncclSuccess && cnt)
        r = g_rccl.AllReduce(cnt, cnt, (size_t)t, ncclUint64, ncclSum, c->comm, (hipStream_t)stream);
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail("mc_allreduce_rank", "ncclAllReduce", r);
    if (e != ncclSuccess) return fail("mc_allreduce_rank", "ncclGroupEnd", e);
    return MC_OK;
}

extern "C" void mc_comm_destroy(mc_comm *c) {
    if (!c) return;
    if (g_ok) g_rccl.CommDestroy(c->comm);
    delete c;
}
