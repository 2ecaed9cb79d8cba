// Threshold enumeration on the GPU (gfx950): every string the model gives -log2 p < threshold, in the
// order the reference's recursive enumerators write them
//     nwords/nwords_enumerator.py:15-42      (fixed order: context = last `order` symbols; a node with
//                                             cur_len > 30 is not expanded; the end symbol never recurses)
//     backwords/backwords_enumerator.py:15-45 (context = longest suffix that is a row; cur_len > 256;
//                                             an end symbol met before min_len is APPENDED and the walk
//                                             goes on from the empty context -- kept, it is what the
//                                             reference writes)
// over the compiled table: payload[e].next already is "the row after taking entry e" under the model's
// own context rule (mc_table.cpp), for the end symbol too.
//
// The recursion becomes a level-synchronous frontier expansion.  Level d holds every node of depth d in
// lexicographic path order (children of one node contiguous and in row order), so the reference's
// DFS pre-order of the written strings is the lexicographic order of the leaves: leaf counts are summed
// bottom-up, rank bases handed out top-down, and every written string lands at its DFS rank.
//   k_enum_expand<false>  per node: entries with acc + ml2p < threshold  -> child count
//   scan                  exclusive prefix of the counts (u64)
//   k_enum_expand<true>   the same walk, children written at their offsets
//   k_enum_leaves / k_enum_ranks / k_enum_emit
// Levels are library-owned device buffers (one arena per level); the host loop reads one u64 per level.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "mc_internal.h"

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            mc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MC_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u32 LEAF = 0xFFFFFFFEu;  // state of a written string (its last transition was the end symbol)

struct Level {  // device arrays of one depth, n nodes
    void *arena = nullptr;
    u64 n = 0;
    u32 *st = nullptr;    // row to expand, LEAF for a written string
    u32 *len = nullptr;   // cur_len (characters)
    u32 *par = nullptr;   // parent's index in the previous level
    u32 *sym = nullptr;   // symbol of the transition that led here
    double *acc = nullptr;
    u64 *off = nullptr;   // [n + 1] first child in the next level
    u64 *leaves = nullptr;
    u64 *rank = nullptr;
};

struct EnumArgs {
    const mc_state_rec *state;
    const mc_payload *payload;
    const u32 *sym_chars;
    double threshold;
    u32 min_len, max_len, backoff, n_ctx;
};

// One lane per node of the level: the loop of `iterate` over next_candidates.items().
template <bool FILL>
__global__ void __launch_bounds__(256) k_enum_expand(EnumArgs a, Level cur, Level nxt, u32 *cnt_out, u32 *bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur.n) return;
    const u32 st = cur.st[i];
    const double acc = cur.acc[i];
    const u32 len = cur.len[i];
    u32 k = 0;
    // (iterate looks the row up BEFORE it tests cur_len / cur_prob: a node without a row is a KeyError
    // in the reference however deep it is -- reported at the node's creation below)
    if (st != LEAF && !(len > a.max_len) && !(acc >= a.threshold)) {
        const mc_state_rec row = a.state[st];
        const u64 o = FILL ? cur.off[i] : 0;
        for (u32 j = 0; j < row.len; ++j) {
            const mc_payload p = a.payload[row.ent_off + j];
            const double na = acc + p.ml2p;
            if (!(na < a.threshold)) continue;
            const bool end = (p.sym & 0x80000000u) != 0;
            const u32 s = p.sym & 0x7FFFFFFFu;
            u32 child;
            if (end && (len >= a.min_len || !a.backoff)) {
                if (len < a.min_len) continue;  // fixed order: too short an end is dropped (:34-39)
                child = LEAF;
            } else {
                child = p.next;  // (backoff: the end symbol met early joins the string, :36-41)
                if (child >= a.n_ctx) {
                    atomicOr(bad, 1u);
                    continue;
                }
            }
            if (FILL) {
                nxt.st[o + k] = child;
                nxt.acc[o + k] = na;
                nxt.len[o + k] = len + a.sym_chars[s];
                nxt.par[o + k] = (u32)i;
                nxt.sym[o + k] = s;
            }
            ++k;
        }
    }
    if (!FILL) cnt_out[i] = k;
}

// exclusive scan of u32 counts into u64 offsets: chunk sums, one workgroup over the chunk sums, apply
constexpr int SC_THREADS = 256, SC_ITEMS = 8, SC_TILE = SC_THREADS * SC_ITEMS, SC_MAX_BLOCKS = 1024;

__device__ __forceinline__ u64 block_excl_scan_u64(u64 v, u64 *wave_tot, u64 &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u64 inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const u64 u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    u64 below = 0, tot = 0;
    for (int w = 0; w < SC_THREADS / 64; ++w) {
        if (w < wid) below += wave_tot[w];
        tot += wave_tot[w];
    }
    total = tot;
    return below + inc - v;
}

__global__ void __launch_bounds__(SC_THREADS) k_scan_sums(const u32 *__restrict__ cnt, u64 n, u64 chunk,
                                                           u64 *__restrict__ partials) {
    __shared__ u64 wave_tot[SC_THREADS / 64];
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    u64 s = 0;
    for (u64 i = lo + threadIdx.x; i < hi; i += SC_THREADS) s += cnt[i];
    u64 tot;
    (void)block_excl_scan_u64(s, wave_tot, tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SC_THREADS) k_scan_partials_u64(u64 *partials, u32 nb, u64 *total) {
    __shared__ u64 wave_tot[SC_THREADS / 64];
    const u32 per = (nb + SC_THREADS - 1) / SC_THREADS;
    const u32 lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    u64 s = 0;
    for (u32 i = lo; i < hi; ++i) s += partials[i];
    u64 tot;
    u64 run = block_excl_scan_u64(s, wave_tot, tot);
    for (u32 i = lo; i < hi; ++i) {
        const u64 v = partials[i];
        partials[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) *total = tot;
}

__global__ void __launch_bounds__(SC_THREADS) k_scan_apply(const u32 *__restrict__ cnt, u64 n, u64 chunk,
                                                            const u64 *__restrict__ partials, u64 *__restrict__ off) {
    __shared__ u64 wave_tot[SC_THREADS / 64];
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    u64 carry = partials[blockIdx.x];
    for (u64 t0 = lo; t0 < hi; t0 += SC_TILE) {
        const u64 i0 = t0 + (u64)threadIdx.x * SC_ITEMS;
        u32 c[SC_ITEMS];
        u64 s = 0;
#pragma unroll
        for (int j = 0; j < SC_ITEMS; ++j) {
            c[j] = i0 + j < hi ? cnt[i0 + j] : 0u;
            s += c[j];
        }
        u64 tot;
        u64 run = carry + block_excl_scan_u64(s, wave_tot, tot);
#pragma unroll
        for (int j = 0; j < SC_ITEMS; ++j) {
            if (i0 + j < hi) off[i0 + j] = run;
            run += c[j];
        }
        carry += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) off[n] = carry;  // (the last chunk ends at n)
}

// written strings below every node (bottom-up) ...
__global__ void __launch_bounds__(256) k_enum_leaves(Level cur, Level nxt, int has_next) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur.n) return;
    u64 s = cur.st[i] == LEAF ? 1 : 0;
    if (has_next)
        for (u64 c = cur.off[i]; c < cur.off[i + 1]; ++c) s += nxt.leaves[c];
    cur.leaves[i] = s;
}

// ... and the DFS rank of the first string below every child (top-down)
__global__ void __launch_bounds__(256) k_enum_ranks(Level cur, Level nxt) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur.n) return;
    u64 run = cur.rank[i];
    for (u64 c = cur.off[i]; c < cur.off[i + 1]; ++c) {
        nxt.rank[c] = run;
        run += nxt.leaves[c];
    }
}

// a written string: its -log2 p and the symbols on the path from the root, at its DFS rank
__global__ void __launch_bounds__(256) k_enum_emit(Level cur, u32 depth, const u32 *const *__restrict__ par_of,
                                                   const u32 *const *__restrict__ sym_of, double *__restrict__ out_ml2p,
                                                   u32 *__restrict__ out_len, u32 *__restrict__ out_syms, u32 stride) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur.n || cur.st[i] != LEAF) return;
    const u64 r = cur.rank[i];
    out_ml2p[r] = cur.acc[i];
    out_len[r] = depth - 1;  // the end symbol that closes the string is not part of it
    u32 at = cur.par[i];
    for (u32 d = depth - 1; d >= 1; --d) {
        out_syms[r * stride + (d - 1)] = sym_of[d][at];
        at = par_of[d][at];
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int alloc_level(Level &L, u64 n) {
    L.n = n;
    const size_t b4 = align256(n * 4), b8 = align256(n * 8), b8p = align256((n + 1) * 8);
    const size_t total = 4 * b4 + 3 * b8 + b8p;
    hipError_t e = hipMalloc(&L.arena, total ? total : 256);
    if (e != hipSuccess) {
        mc_set_error("mc_enumerate: hipMalloc(%zu) for a level of %llu nodes failed: %s", total, (u64)n, hipGetErrorString(e));
        L.arena = nullptr;
        return MC_ERR_NOMEM;
    }
    char *p = (char *)L.arena;
    L.st = (u32 *)p; p += b4;
    L.len = (u32 *)p; p += b4;
    L.par = (u32 *)p; p += b4;
    L.sym = (u32 *)p; p += b4;
    L.acc = (double *)p; p += b8;
    L.leaves = (u64 *)p; p += b8;
    L.rank = (u64 *)p; p += b8;
    L.off = (u64 *)p;
    return MC_OK;
}

u32 blocks_for(u64 n) { return (u32)((n + 255) / 256); }

}  // namespace

struct mc_enum_result {
    int device = 0;
    u64 n_out = 0, n_nodes = 0;
    u32 stride = 0, levels = 0;
    double *ml2p = nullptr;
    u32 *len = nullptr, *syms = nullptr;
};

extern "C" void mc_enum_free(mc_enum_result *r) {
    if (!r) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(r->device);
    if (r->ml2p) (void)hipFree(r->ml2p);
    if (r->len) (void)hipFree(r->len);
    if (r->syms) (void)hipFree(r->syms);
    (void)hipSetDevice(prev);
    delete r;
}

extern "C" int mc_enum_info(const mc_enum_result *r, uint64_t info[4]) {
    if (!r || !info) return MC_ERR_ARG;
    info[0] = r->n_out;
    info[1] = r->stride;
    info[2] = r->n_nodes;
    info[3] = r->levels;
    return MC_OK;
}

extern "C" int64_t mc_enum_fetch(const mc_enum_result *r, int what, void *dst, size_t cap) {
    if (!r) return MC_ERR_ARG;
    const void *src;
    size_t bytes;
    switch (what) {
    case MC_ENUM_ML2P: src = r->ml2p; bytes = r->n_out * sizeof(double); break;
    case MC_ENUM_LEN: src = r->len; bytes = r->n_out * sizeof(u32); break;
    case MC_ENUM_SYMS: src = r->syms; bytes = r->n_out * (size_t)r->stride * sizeof(u32); break;
    default: mc_set_error("mc_enum_fetch: unknown array %d", what); return MC_ERR_ARG;
    }
    if (dst && cap && bytes) {
        int prev = 0;
        HIP_TRY(hipGetDevice(&prev));
        HIP_TRY(hipSetDevice(r->device));
        HIP_TRY(hipMemcpy(dst, src, cap < bytes ? cap : bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipSetDevice(prev));
    }
    return (int64_t)bytes;
}

extern "C" int mc_enumerate(const mc_model *m, double threshold, uint32_t min_len, uint32_t max_len, uint64_t max_nodes,
                            void *stream, mc_enum_result **out) {
    if (!m || !out) {
        mc_set_error("mc_enumerate: null argument");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    mc_enum_view v;
    int rc = mc_model_enum_view(m, &v);
    if (rc != MC_OK) return rc;
    if (v.kind != MC_KIND_NGRAM && v.kind != MC_KIND_BACKOFF) {
        mc_set_error("mc_enumerate: the reference enumerates fixed-order and backoff models only");
        return MC_ERR_ARG;
    }
    if (v.start_state >= v.n_ctx) {
        mc_set_error("mc_enumerate: the start context has no row (KeyError in the reference, nwords_enumerator.py:26)");
        return MC_ERR_TABLE;
    }
    hipStream_t s = (hipStream_t)stream;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(v.device));
    EnumArgs a{v.state, v.payload, v.sym_chars, threshold, min_len, max_len, v.kind == MC_KIND_BACKOFF ? 1u : 0u, v.n_ctx};

    std::vector<Level> lv;
    u32 *cnt = nullptr, *bad = nullptr;
    u64 *partials = nullptr, *total_d = nullptr;
    const u32 **par_of = nullptr, **sym_of = nullptr;
    mc_enum_result *res = nullptr;
    auto cleanup = [&]() {
        for (Level &L : lv)
            if (L.arena) (void)hipFree(L.arena);
        if (cnt) (void)hipFree(cnt);
        if (partials) (void)hipFree(partials);  // (bad and total_d live in the same allocation)
        if (par_of) (void)hipFree((void *)par_of);
        (void)hipSetDevice(prev);
    };
#define ENUM_TRY(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            mc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);         \
            cleanup();                                                                                       \
            if (res) mc_enum_free(res);                                                                      \
            return MC_ERR_HIP;                                                                               \
        }                                                                                                    \
    } while (0)

    ENUM_TRY(hipMalloc((void **)&partials, (SC_MAX_BLOCKS + 2) * sizeof(u64)));
    total_d = partials + SC_MAX_BLOCKS;
    bad = (u32 *)(partials + SC_MAX_BLOCKS + 1);
    ENUM_TRY(hipMemsetAsync(bad, 0, sizeof(u64), s));

    // level 0: the empty string in the start context (cur_pwd = the start symbols, cur_prob = .0)
    lv.emplace_back();
    if ((rc = alloc_level(lv[0], 1)) != MC_OK) {
        cleanup();
        return rc;
    }
    {
        const u32 st0 = v.start_state, zero = 0, none = MC_NONE;
        const double acc0 = 0.0;
        ENUM_TRY(hipMemcpyAsync(lv[0].st, &st0, 4, hipMemcpyHostToDevice, s));
        ENUM_TRY(hipMemcpyAsync(lv[0].len, &zero, 4, hipMemcpyHostToDevice, s));
        ENUM_TRY(hipMemcpyAsync(lv[0].par, &none, 4, hipMemcpyHostToDevice, s));
        ENUM_TRY(hipMemcpyAsync(lv[0].sym, &none, 4, hipMemcpyHostToDevice, s));
        ENUM_TRY(hipMemcpyAsync(lv[0].acc, &acc0, 8, hipMemcpyHostToDevice, s));
        ENUM_TRY(hipStreamSynchronize(s));
    }
    u64 n_nodes = 1, cnt_cap = 0;
    for (;;) {
        Level &cur = lv.back();
        if (cur.n > cnt_cap) {
            if (cnt) (void)hipFree(cnt);
            cnt = nullptr;
            cnt_cap = cur.n + cur.n / 2;
            ENUM_TRY(hipMalloc((void **)&cnt, cnt_cap * sizeof(u32)));
        }
        Level none;
        hipLaunchKernelGGL(k_enum_expand<false>, dim3(blocks_for(cur.n)), dim3(256), 0, s, a, cur, none, cnt, bad);
        const u64 tiles = (cur.n + SC_TILE - 1) / SC_TILE;
        const u32 sb = (u32)(tiles < (u64)SC_MAX_BLOCKS ? tiles : SC_MAX_BLOCKS);
        const u64 chunk = ((tiles + sb - 1) / sb) * SC_TILE;
        const u32 used = (u32)((cur.n + chunk - 1) / chunk);  // chunks that hold nodes (the last one ends at n)
        hipLaunchKernelGGL(k_scan_sums, dim3(used), dim3(SC_THREADS), 0, s, (const u32 *)cnt, cur.n, chunk, partials);
        hipLaunchKernelGGL(k_scan_partials_u64, dim3(1), dim3(SC_THREADS), 0, s, partials, used, total_d);
        hipLaunchKernelGGL(k_scan_apply, dim3(used), dim3(SC_THREADS), 0, s, (const u32 *)cnt, cur.n, chunk,
                           (const u64 *)partials, cur.off);
        u64 host[2] = {0, 0};  // children in total, error word
        ENUM_TRY(hipMemcpyAsync(host, total_d, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
        ENUM_TRY(hipStreamSynchronize(s));
        if (host[1] & 0xFFFFFFFFull) {
            mc_set_error("mc_enumerate: a string below the threshold leads to a context without a row (KeyError in "
                         "the reference, nwords_enumerator.py:26)");
            cleanup();
            return MC_ERR_TABLE;
        }
        const u64 kids = host[0];
        if (kids == 0) break;
        if (kids > 0xFFFFFFF0ull || n_nodes + kids > max_nodes) {
            mc_set_error("mc_enumerate: depth %zu would hold %llu nodes (%llu so far, limit %llu): lower the threshold "
                         "or raise max_nodes", lv.size(), (u64)kids, (u64)n_nodes, (u64)max_nodes);
            cleanup();
            return MC_ERR_NOMEM;
        }
        lv.emplace_back();
        if ((rc = alloc_level(lv.back(), kids)) != MC_OK) {
            cleanup();
            return rc;
        }
        Level &par = lv[lv.size() - 2];
        hipLaunchKernelGGL(k_enum_expand<true>, dim3(blocks_for(par.n)), dim3(256), 0, s, a, par, lv.back(), cnt, bad);
        n_nodes += kids;
    }
    const u32 depth_n = (u32)lv.size();
    // leaf counts bottom-up, rank bases top-down
    for (int d = (int)depth_n - 1; d >= 0; --d) {
        const int has_next = d + 1 < (int)depth_n;
        hipLaunchKernelGGL(k_enum_leaves, dim3(blocks_for(lv[d].n)), dim3(256), 0, s, lv[d], has_next ? lv[d + 1] : Level(),
                           has_next);
    }
    ENUM_TRY(hipMemsetAsync(lv[0].rank, 0, sizeof(u64), s));
    for (u32 d = 0; d + 1 < depth_n; ++d)
        hipLaunchKernelGGL(k_enum_ranks, dim3(blocks_for(lv[d].n)), dim3(256), 0, s, lv[d], lv[d + 1]);
    u64 n_out = 0;
    ENUM_TRY(hipMemcpyAsync(&n_out, lv[0].leaves, sizeof(u64), hipMemcpyDeviceToHost, s));
    ENUM_TRY(hipStreamSynchronize(s));

    res = new (std::nothrow) mc_enum_result();
    if (!res) {
        cleanup();
        return MC_ERR_NOMEM;
    }
    res->device = v.device;
    res->n_out = n_out;
    res->n_nodes = n_nodes;
    res->levels = depth_n;
    res->stride = depth_n > 2 ? depth_n - 2 : 1;  // symbols of the longest written string
    if (n_out) {
        ENUM_TRY(hipMalloc((void **)&res->ml2p, n_out * sizeof(double)));
        ENUM_TRY(hipMalloc((void **)&res->len, n_out * sizeof(u32)));
        ENUM_TRY(hipMalloc((void **)&res->syms, n_out * (size_t)res->stride * sizeof(u32)));
        ENUM_TRY(hipMemsetAsync(res->syms, 0, n_out * (size_t)res->stride * sizeof(u32), s));
        std::vector<const u32 *> hp(2 * (size_t)depth_n);
        for (u32 d = 0; d < depth_n; ++d) {
            hp[d] = lv[d].par;
            hp[depth_n + d] = lv[d].sym;
        }
        ENUM_TRY(hipMalloc((void **)&par_of, hp.size() * sizeof(u32 *)));
        sym_of = par_of + depth_n;
        ENUM_TRY(hipMemcpyAsync((void *)par_of, hp.data(), hp.size() * sizeof(u32 *), hipMemcpyHostToDevice, s));
        for (u32 d = 1; d < depth_n; ++d)
            hipLaunchKernelGGL(k_enum_emit, dim3(blocks_for(lv[d].n)), dim3(256), 0, s, lv[d], d, (const u32 *const *)par_of,
                               (const u32 *const *)sym_of, res->ml2p, res->len, res->syms, res->stride);
        ENUM_TRY(hipStreamSynchronize(s));
    }
    ENUM_TRY(hipGetLastError());
    cleanup();
    *out = res;
    return MC_OK;
#undef ENUM_TRY
}
