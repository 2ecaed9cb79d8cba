// Run-length rank index (gfx950): MonteCarloLib.__init__ + ml2p2rank, reference lib4mc/MonteCarloLib.py:10-32,
// built from what the query actually reads.
//
//     keys.sort(); positions = cumsum(2 ** (keys - log2 n)); idx = bisect_right(keys, x); positions[idx - 1]
//
// bisect_right lands on the END of a run of equal keys, always: of the n sorted keys and n positions the reference
// materialises, a query only ever reads one entry per DISTINCT value.  Sampled -log2 p repeat heavily (headline
// model: 1e7 samples, 170,340 distinct values, 76 % of the samples copies of 240 of them), so the index kept here
// is one entry per distinct value v, ascending:
//     vals[d]       the value
//     cum_count[d]  samples <= v                    (what bisect_right returns)
//     cum_pos[d]    sum of 2^(key - log2 n) over them  (what positions[idx - 1] holds)
// and the sorted array itself is never written (mc_rank_runs_expand writes it for callers that ask: the reference
// sorts the caller's list in place).
//
//   k_rl_sniff   one workgroup: a 16384-key sample gives the values that repeat often ("heavy", <= 1024, ascending)
//                and the range [lo, hi] the other keys spread over.  The list and the range of the last build of n
//                keys in the same workspace are taken again while they still fit (any list and any range give a
//                correct index; stale ones only cost time).
//   k_rl_strip   ONE read of all keys.  A copy of a heavy value is counted in LDS.  Every other key ("tail") goes to
//                an LDS buffer; a full buffer (4096 keys, a "batch") is ordered by BIN -- bin = floor((key - lo) *
//                P / (hi - lo)), clamped: monotone in the key -- and written out with the offsets of its P + 1 bin
//                borders.  No global atomics per key, nothing another workgroup has to wait for.
//   k_rl_bins    one workgroup per bin: collects the bin's segment of every batch and the heavy values that fall
//                into the bin, counts them in an LDS hash table, sorts the distinct values (rank by counting) and
//                writes the bin's runs with their running counts and weights.  Bins are ordered, so the runs of
//                bin b all lie below those of bin b + 1.
//   k_rl_final   prefix over the bins, runs copied to their final places.
//
// A device-wide hash table of the distinct values was measured first and dropped: scattered device-scope atomics
// run at ~2e10 a second on this part (scripts/ubench/global_atomics.hip: 2e6 CAS + add pairs take 105 us), the 1.9e6
// tail keys of the headline step would cost more than the whole sort does now.  Plain stores of the same keys take 11 us.
//
// Limits: n <= 2^26 keys and at most CAPD distinct values per bin.  A bin that holds more (keys that do not
// look like a model's scores at all: uniform random numbers, millions of distinct values) sets the error word;
// the index is then marked failed (queries answer NaN) and the caller falls back to mc_rank_build, which sorts
// anything.  Nothing is ever written outside the buffers.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>

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

constexpr int WG = 512;            // threads of the strip / bin / final workgroups
constexpr int TILE = 4096;         // keys per tile of k_rl_strip
constexpr int ITEMS = TILE / WG;   // keys per lane and tile
constexpr int NWAVE = WG / 64;
constexpr int STRIP_SLOTS = 512;   // k_rl_strip workgroups the GPU holds at once (two a CU): regions are sized for one round
constexpr int H_MAX = 1024, HSET = 2048;
constexpr int SNIFF_S = 16384, SNIFF_SLOTS = 16384, SNIFF_MIN = 5, SNIFF_THREADS = 1024, SNIFF_PROBES = 8;
constexpr int SNIFF_BLOCK = 16;    // consecutive keys per sampled block
constexpr int TBUF = TILE;         // tail keys per batch (a tile adds at most TILE keys: one flush per tile at most)
constexpr int P_MAX = 1024;        // bins at most
constexpr int BSLOTS = 4096;       // slots of a bin's counting table
constexpr int CAPD = 2048;         // distinct values a bin may hold
constexpr u64 EMPTY = ~0ull;
constexpr u64 RUNS_MAX_N = 1ull << 26;
constexpr u64 F64_INF = 0x7FF0000000000000ull;
constexpr u32 LIST_MAGIC = 0x4D43524Cu, LIST_MAX_AGE = 63;

enum { ERR_BIN = 1u, ERR_RANGE = 2u, ERR_COUNT = 4u, ERR_DEBUG = 16u };

struct RCtl {
    u32 err;
    u32 n_heavy;
    u32 n_tail;
    u32 n_allones;  // keys equal to the tables' empty marker (a NaN payload): counted apart, the last run
    // the heavy values and the range of one build serve the next builds of as many keys in the same workspace
    u32 list_magic, list_n, list_tail, list_age;
    double map_lo, map_hi;
};

struct RWs {
    RCtl *ctl;
    u64 *heavy;      // [H_MAX] ascending
    u32 *hcnt;       // [H_MAX] their copies
    const u64 *in;
    u64 *tail;       // [n] region r's batches at r * region + j * TBUF, each ordered by bin
    u32 *nbatch;     // [regions] batches region r wrote
    u32 *start;      // [regions * nb][P + 1] offsets of the bins inside a batch
    u64 *rval;       // [P * CAPD] a bin's distinct values, ascending
    u64 *rccum;      // [P * CAPD] ... copies up to and including each, inside the bin
    double *rwcum;   // [P * CAPD] ... and their weights
    u32 *bin_d;      // [P_MAX] distinct values of the bin
    u64 *bin_c;      // [P_MAX] keys of the bin
    double *bin_w;   // [P_MAX] their weights
    u32 n, region, regions, nb, P;
    double logn;
    u32 reuse_list, debug_fail;
    // outputs (the caller's)
    u64 *out_vals, *out_ccum;
    double *out_wcum;
    u64 *out_info;   // {distinct values, error word, keys counted, heavy values, tail keys, builds the list has served, 0, 0}
    u64 out_cap;
};

// ---- small helpers ----------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wg_excl_scan(u32 v, u32 *wtot, u32 &total) {  // every thread calls; wtot: NWAVE words
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    u32 off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) {
        const u32 t = wtot[i];
        if (i < wid) off += t;
        tot += t;
    }
    total = tot;
    return off + inc - v;
}

// exclusive prefix over the workgroup, in thread order, the same additions in every workgroup that scans the same
// values (the bins' bases must agree between the workgroups of k_rl_final).  The exclusive value is shuffled, never
// formed as inclusive - own (weights span 2^-20 .. 2^90).
template <typename T>
__device__ __forceinline__ T wg_excl_scan_t(T v, T *wtot, T &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    T excl = __shfl_up(inc, 1);
    if (lane == 0) excl = (T)0;
    __syncthreads();
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    T off = (T)0, tot = (T)0;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) {
        const T t = wtot[i];
        if (i < wid) off += t;
        tot += t;
    }
    total = tot;
    return off + excl;
}

__device__ __forceinline__ u64 wave_min(u64 v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const u64 t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ u64 wave_max(u64 v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const u64 t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

__device__ __forceinline__ u32 mix32(u64 k) {
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    u32 x = lo ^ (hi * 0x9E3779B1u);
    x ^= x >> 15;
    return x * 0x85EBCA6Bu;
}
__device__ __forceinline__ u32 strip_hash(u64 k) { return mix32(k) >> 21; }  // 11 bits = HSET slots

// The bin of a key: monotone in the key's BIT pattern (the order the index is kept in; for the non-negative doubles
// it is made for, the numeric order): below lo -> 0, at or above hi (infinities, NaNs, a set sign bit) -> P - 1.
struct BinMap {
    u64 lo_bits, hi_bits;
    double lo, scale;
    u32 P;
};
__device__ __forceinline__ BinMap bin_map(const RCtl *c, u32 P) {
    BinMap m;
    m.lo = c->map_lo;
    const double hi = c->map_hi;
    m.lo_bits = (u64)__double_as_longlong(m.lo);
    m.hi_bits = (u64)__double_as_longlong(hi);
    m.scale = hi > m.lo ? (double)P / (hi - m.lo) : 0.0;
    m.P = P;
    return m;
}
__device__ __forceinline__ u32 bin_of(const BinMap &m, u64 kb) {
    if (kb >= m.hi_bits) return m.P - 1;
    if (kb <= m.lo_bits) return 0u;
    const double x = (__longlong_as_double((long long)kb) - m.lo) * m.scale;
    const u32 b = (u32)x;
    return b < m.P ? b : m.P - 1;
}

// ---- the sample: heavy values and the range of the rest ------------------------------------------------------
__device__ __forceinline__ bool sample_index(u32 n, u32 q, u64 &idx) {
    if (n <= (u32)SNIFF_S) {
        idx = q;
        return q < n;
    }
    idx = (u64)(q / SNIFF_BLOCK) * (n / (SNIFF_S / SNIFF_BLOCK)) + (q % SNIFF_BLOCK);  // (blocks of 16: 1024 pages)
    return true;
}

__global__ void __launch_bounds__(SNIFF_THREADS) k_rl_sniff(RWs w) {
    constexpr int PER = SNIFF_S / SNIFF_THREADS;
    static_assert(SNIFF_SLOTS == 16384, "slot = top 14 bits of the mix");
    __shared__ u32 tag[SNIFF_SLOTS];     // 0 = empty
    __shared__ u32 tc[SNIFF_SLOTS / 2];  // two 16-bit counts per word
    __shared__ u32 claim[SNIFF_SLOTS / 32];
    __shared__ u64 list[H_MAX];
    __shared__ u64 red[2 * (SNIFF_THREADS / 64)];
    __shared__ u32 nlist, n_rest;
    __shared__ u32 keep_h, keep_tail, keep_age;
    const u32 n = w.n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        const RCtl c = *w.ctl;
        const u32 base = c.list_age == 0 ? c.n_tail : c.list_tail;  // (the list's own build has told its tail by now)
        const bool ok = w.reuse_list && c.list_magic == LIST_MAGIC && c.list_n == n && c.err == 0 &&
                        c.n_heavy <= (u32)H_MAX && c.list_age < LIST_MAX_AGE && c.n_tail <= n &&
                        (u64)c.n_tail * 4 <= (u64)base * 5 + n / 32 && c.map_hi > c.map_lo;
        keep_h = ok ? c.n_heavy : 0xFFFFFFFFu;  // (a list of no heavy values is a list too)
        keep_tail = base;
        keep_age = c.list_age + 1;
        n_rest = 0;
    }
    __syncthreads();
    u32 keep = keep_h;
    if (keep != 0xFFFFFFFFu) {
        int unordered = 0;
        for (u32 i = tid + 1; i < keep; i += SNIFF_THREADS) unordered |= !(w.heavy[i - 1] < w.heavy[i]);
        if (__syncthreads_or(unordered)) keep = 0xFFFFFFFFu;  // (never a list of ours)
    }
    // what the build counts into starts at zero: this workgroup runs alone and first
    for (u32 i = tid; i < (u32)H_MAX; i += SNIFF_THREADS) w.hcnt[i] = 0;
    if (keep != 0xFFFFFFFFu) {
        __syncthreads();
        if (tid == 0) {
            w.ctl->err = w.debug_fail ? ERR_DEBUG : 0u;
            w.ctl->n_tail = 0;
            w.ctl->n_allones = 0;
            w.ctl->list_tail = keep_tail;
            w.ctl->list_age = keep_age;
        }
        return;
    }
    for (int i = tid; i < SNIFF_SLOTS; i += SNIFF_THREADS) tag[i] = 0;
    for (int i = tid; i < SNIFF_SLOTS / 2; i += SNIFF_THREADS) tc[i] = 0;
    for (int i = tid; i < SNIFF_SLOTS / 32; i += SNIFF_THREADS) claim[i] = 0;
    __syncthreads();
    u64 k[PER];
    u32 slot[PER];
    bool have[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        u64 idx;
        have[q] = sample_index(n, (u32)(q * SNIFF_THREADS + tid), idx);
        k[q] = have[q] ? w.in[idx] : EMPTY;
        slot[q] = 0xFFFFFFFFu;
    }
    const bool look = n >= (u32)SNIFF_S * 4;  // small inputs: no heavy values, the range only
    u32 bar = SNIFF_MIN;
    if (look) {
        // every key tries its home slot first, all PER attempts in flight together; only a key that finds another
        // value there walks on, and gives up after SNIFF_PROBES slots (with few repeats the table fills up)
        u32 tg[PER], old[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 x = mix32(k[q]);
            tg[q] = x | 1u;
            slot[q] = x >> 18;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) old[q] = atomicCAS(&tag[slot[q]], 0u, tg[q]);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            u32 s = slot[q];
            bool placed = old[q] == 0u || old[q] == tg[q];
            for (int probe = 1; !placed && probe < SNIFF_PROBES; ++probe) {
                s = (s + 1) & (SNIFF_SLOTS - 1);
                const u32 o = atomicCAS(&tag[s], 0u, tg[q]);
                placed = o == 0u || o == tg[q];
            }
            slot[q] = placed && k[q] != EMPTY ? s : 0xFFFFFFFFu;  // (the empty marker is never heavy)
        }
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (slot[q] != 0xFFFFFFFFu) atomicAdd(&tc[slot[q] >> 1], 1u << (16 * (slot[q] & 1u)));
        for (;;) {  // (uniform) the lowest bar that leaves at most H_MAX values
            __syncthreads();
            if (tid == 0) nlist = 0;
            __syncthreads();
            u32 mine = 0;
            for (int i = tid; i < SNIFF_SLOTS; i += SNIFF_THREADS) mine += ((tc[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) >= bar;
            if (mine) atomicAdd(&nlist, mine);
            __syncthreads();
            if (nlist <= (u32)H_MAX) break;
            ++bar;
        }
    }
    __syncthreads();
    if (tid == 0) nlist = 0;
    __syncthreads();
    bool over[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 s = slot[q];
        over[q] = look && s != 0xFFFFFFFFu && ((tc[s >> 1] >> (16 * (s & 1u))) & 0xFFFFu) >= bar;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (over[q]) {
            const u32 s = slot[q], bit = 1u << (s & 31u);
            if (!(atomicOr(&claim[s >> 5], bit) & bit)) {
                const u32 p = atomicAdd(&nlist, 1u);
                list[p] = k[q];  // one claim per slot, at most H_MAX slots at the bar
            }
        }
    }
    // the range of the keys that are not heavy (finite ones; all finite sample keys when those are too few to tell)
    for (int pass = 0; pass < 2; ++pass) {
        u64 mn = EMPTY, mx = 0;
        u32 cnt = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (have[q] && k[q] < F64_INF && (pass == 1 || !over[q])) {
                mn = k[q] < mn ? k[q] : mn;
                mx = k[q] > mx ? k[q] : mx;
                ++cnt;
            }
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
#pragma unroll
        for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
        __syncthreads();
        if (lane == 0) {
            red[wid] = mn;
            red[SNIFF_THREADS / 64 + wid] = mx;
            if (cnt) atomicAdd(&n_rest, cnt);
        }
        __syncthreads();
        if (n_rest >= 64u || pass == 1) break;  // (uniform)
        __syncthreads();
        if (tid == 0) n_rest = 0;
        __syncthreads();
    }
    __syncthreads();
    const u32 H = nlist;
    {   // the values are distinct: rank = position
        u32 *rank = tc;  // (the counts are not needed any more)
        for (u32 i = tid; i < H; i += SNIFF_THREADS) rank[i] = 0;
        __syncthreads();
        u32 parts = 1;
        while (parts * 2 * H <= (u32)SNIFF_THREADS && parts < 16) parts *= 2;
        const u32 i = tid / parts, part = tid % parts;
        if (i < H) {
            const u64 v = list[i];
            const u32 per = (H + parts - 1) / parts, j0 = part * per, j1 = j0 + per < H ? j0 + per : H;
            u32 r = 0;
#pragma unroll 8
            for (u32 j = j0; j < j1; ++j) r += list[j] < v ? 1u : 0u;
            if (r) atomicAdd(&rank[i], r);
        }
        __syncthreads();
        for (u32 i2 = tid; i2 < H; i2 += SNIFF_THREADS) w.heavy[rank[i2]] = list[i2];
    }
    if (tid == 0) {
        u64 mn = EMPTY, mx = 0;
        for (int i = 0; i < SNIFF_THREADS / 64; ++i) {
            mn = red[i] < mn ? red[i] : mn;
            mx = red[SNIFF_THREADS / 64 + i] > mx ? red[SNIFF_THREADS / 64 + i] : mx;
        }
        double lo = 0.0, hi = 1.0;
        if (mn <= mx && mn != EMPTY) {
            lo = __longlong_as_double((long long)mn);
            hi = __longlong_as_double((long long)mx);
            // a sample sees the middle of the distribution: the range is widened so that the ends, which only the
            // full input reaches, spread over bins of their own instead of piling up in the first and the last
            const double range = hi - lo;
            lo = lo - 0.10 * range;
            if (lo < 0.0) lo = 0.0;
            hi = hi + 0.25 * range;
            if (!(hi > lo)) hi = lo + 1.0;
            if (!(hi < __longlong_as_double((long long)F64_INF))) hi = __longlong_as_double((long long)(F64_INF - 1));
        }
        RCtl *c = w.ctl;
        c->err = w.debug_fail ? ERR_DEBUG : 0u;
        c->n_heavy = H;
        c->n_tail = 0;
        c->n_allones = 0;
        c->list_magic = LIST_MAGIC;  // (list_age = 0: the next build reads this build's n_tail)
        c->list_n = n;
        c->list_tail = 0;
        c->list_age = 0;
        c->map_lo = lo;
        c->map_hi = hi;
    }
}

// ---- strip: count the copies of heavy values, order the rest by bin ------------------------------------------
__global__ void __launch_bounds__(WG) k_rl_strip(RWs w) {
    static_assert(HSET == 2048, "strip_hash yields 11 bits");
    static_assert(TBUF == TILE && TBUF == ITEMS * WG, "a batch is flushed by the whole workgroup, ITEMS keys a lane");
    static_assert(P_MAX == 2 * WG, "the scan of the bins' counts takes two bins a thread");
    __shared__ u64 hk[HSET + 1];   // (+1: the pair read of the last slot)
    __shared__ u32 hs[HSET + 1];   // copies counted per slot
    __shared__ u32 hi[HSET];       // slot -> index of the heavy value
    __shared__ u64 tb[TBUF];       // tail keys waiting for their batch
    __shared__ u32 bh[P_MAX + 1];  // keys per bin of the batch, then the bins' offsets
    __shared__ u32 wtot[NWAVE];
    __shared__ u32 tfill;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 H = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
    const u32 n = w.n, P = w.P;
    const BinMap bm = bin_map(w.ctl, P);
    for (u32 i = tid; i <= (u32)HSET; i += WG) {
        hk[i] = EMPTY;
        hs[i] = 0;
    }
    __syncthreads();
    for (u32 i = tid; i < H; i += WG) {
        const u64 v = w.heavy[i];
        if (v == EMPTY) continue;
        u32 s = strip_hash(v);
        for (;;) {  // H <= HSET / 2 distinct values
            const u64 old = atomicCAS(&hk[s], EMPTY, v);
            if (old == EMPTY) {
                hi[s] = i;
                break;
            }
            if (old == v) break;  // (a list that repeats a value: never ours, but never a hang either)
            s = (s + 1) & (HSET - 1);
        }
    }
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
    u32 tail_total = 0, allones = 0;  // (thread 0 / every thread)

    // one batch: the first m keys of tb, ordered by bin, to `out`; the bins' offsets to `row`
    auto flush = [&](u32 m, u64 *out, u32 room, u32 *row) {
        for (u32 i = tid; i <= (u32)P_MAX; i += WG) bh[i] = 0;
        __syncthreads();
        u64 key[ITEMS];
        u32 bin[ITEMS], rk[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            key[q] = i < m ? tb[i] : 0ull;
            bin[q] = bin_of(bm, key[q]);
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {  // (independent returning adds, all in flight together)
            const u32 i = q * WG + tid;
            rk[q] = i < m ? atomicAdd(&bh[bin[q]], 1u) : 0u;
        }
        __syncthreads();
        const u32 c0 = bh[2 * tid], c1 = bh[2 * tid + 1];
        u32 tot;
        const u32 ex = wg_excl_scan(c0 + c1, wtot, tot);
        bh[2 * tid] = ex;
        bh[2 * tid + 1] = ex + c0;
        if (tid == 0) bh[P_MAX] = m;
        __syncthreads();
        for (u32 i = tid; i <= P; i += WG) row[i] = bh[i];  // (bins >= P are empty: bh[P] = m)
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            if (i < m) {
                const u32 at = bh[bin[q]] + rk[q];
                if (at < room) out[at] = key[q]; else atomicOr(&w.ctl->err, ERR_RANGE);
            }
        }
        __syncthreads();
    };

    for (u32 r = blockIdx.x; r < w.regions; r += gridDim.x) {
        const u64 lo = (u64)r * w.region;
        const u32 rn = (u32)(n - lo < (u64)w.region ? n - lo : (u64)w.region);
        u32 nbat = 0;
        if (tid == 0) tfill = 0;
        __syncthreads();
        u64 k[ITEMS], nk[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 j = q * WG + tid;
            nk[q] = w.in[lo + (j < rn ? j : rn - 1)];  // (clamped, not guarded: no branch; rn >= 1)
        }
        for (u32 c0 = 0; c0 < rn; c0 += TILE) {
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) k[q] = nk[q];
            if (c0 + TILE < rn) {  // the next tile travels while this one is looked up
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = c0 + TILE + q * WG + tid;
                    nk[q] = w.in[lo + (j < rn ? j : rn - 1)];
                }
            }
            // heavy test without branches: the key's home slot and the next one are read together; only a key that
            // meets two other values there walks on (rare: the table is at most half full)
            u64 tmask[ITEMS];
            u32 total = 0;
            {
                u32 s[ITEMS];
                u64 a[ITEMS], b[ITEMS];
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    s[q] = strip_hash(k[q]);
                    a[q] = hk[s[q]];
                    b[q] = hk[s[q] + 1];  // (slot HSET is a spare that stays empty)
                }
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const bool valid = c0 + q * WG + tid < rn;
                    const bool real = valid && k[q] != EMPTY;
                    if (valid && !real) ++allones;
                    const bool hit_a = real && a[q] == k[q];
                    const bool hit_b = real && !hit_a && a[q] != EMPTY && b[q] == k[q];
                    bool walk = real && !hit_a && !hit_b && a[q] != EMPTY && (b[q] != EMPTY || s[q] + 1 == (u32)HSET);
                    bool heavy = hit_a || hit_b;
                    u32 slot = s[q] + (hit_b ? 1u : 0u);
                    if (__ballot(walk)) {  // (rare)
                        if (walk) {
                            u32 t = s[q] + 1 == (u32)HSET ? 0u : (s[q] + 2 == (u32)HSET ? 0u : s[q] + 2);
                            for (;;) {
                                const u64 v = hk[t];
                                if (v == k[q]) {
                                    heavy = true;
                                    slot = t;
                                    break;
                                }
                                if (v == EMPTY) break;
                                t = t + 1 == (u32)HSET ? 0u : t + 1;
                            }
                        }
                    }
                    if (heavy) atomicAdd(&hs[slot], 1u);
                    tmask[q] = __ballot(real && !heavy);
                    total += (u32)__popcll(tmask[q]);
                }
            }
            // the tail keys of the tile take their places in the buffer from ONE add per wave
            u32 base = 0;
            if (total) {
                if (lane == 0) base = atomicAdd(&tfill, total);
                base = __builtin_amdgcn_readfirstlane(base);
            }
            u32 pos[ITEMS];
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const bool tail = (tmask[q] >> lane) & 1ull;
                pos[q] = tail ? base + (u32)__popcll(tmask[q] & lt) : 0xFFFFFFFFu;
                if (tail && pos[q] < (u32)TBUF) tb[pos[q]] = k[q];
                base += (u32)__popcll(tmask[q]);
            }
            __syncthreads();
            if (tfill >= (u32)TBUF) {  // (uniform) a full batch; what did not fit opens the next one
                const u32 at = nbat * (u32)TBUF;
                flush((u32)TBUF, w.tail + lo + at, at < rn ? rn - at : 0u,
                      w.start + ((size_t)r * w.nb + (nbat < w.nb ? nbat : w.nb - 1)) * (P + 1));
                if (nbat >= w.nb && tid == 0) atomicOr(&w.ctl->err, ERR_RANGE);  // (more tail keys than keys: impossible)
                ++nbat;
#pragma unroll
                for (int q = 0; q < ITEMS; ++q)
                    if (pos[q] != 0xFFFFFFFFu && pos[q] >= (u32)TBUF) tb[pos[q] - (u32)TBUF] = k[q];
                if (tid == 0) {
                    tfill -= (u32)TBUF;
                    tail_total += (u32)TBUF;
                }
                __syncthreads();
            }
        }
        const u32 m = tfill;
        __syncthreads();
        if (m) {
            const u32 at = nbat * (u32)TBUF;
            flush(m, w.tail + lo + at, at < rn ? rn - at : 0u,
                  w.start + ((size_t)r * w.nb + (nbat < w.nb ? nbat : w.nb - 1)) * (P + 1));
            if (nbat >= w.nb && tid == 0) atomicOr(&w.ctl->err, ERR_RANGE);
            ++nbat;
            if (tid == 0) tail_total += m;
        }
        if (tid == 0) w.nbatch[r] = nbat < w.nb ? nbat : w.nb;
        __syncthreads();
    }
    if (tid == 0 && tail_total) atomicAdd(&w.ctl->n_tail, tail_total);
#pragma unroll
    for (int o = 32; o; o >>= 1) allones += __shfl_xor(allones, o);
    if (lane == 0 && allones) atomicAdd(&w.ctl->n_allones, allones);
    // per-slot counts -> per-value counts in LDS, then ONE coalesced add per value (adds scattered over the array
    // in slot order are one memory-side request each and queue on its few lines)
    __syncthreads();
    static_assert(TBUF * 2 >= H_MAX, "tb is reused for the per-value counts");
    u32 *hc = (u32 *)tb;
    for (u32 i = tid; i < H; i += WG) hc[i] = 0;
    __syncthreads();
    for (u32 i = tid; i < (u32)HSET; i += WG)
        if (hk[i] != EMPTY && hs[i]) hc[hi[i]] = hs[i];
    __syncthreads();
    for (u32 i = tid; i < H; i += WG)
        if (hc[i]) atomicAdd(&w.hcnt[i], hc[i]);
}

// ---- one bin: count, sort, scan ------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_rl_bins(RWs w) {
    __shared__ u64 tk[BSLOTS];   // the counting table, then the sorted values
    __shared__ u32 tc[BSLOTS];   // ... their counts
    __shared__ u64 ev[CAPD];     // the table's entries, compacted
    __shared__ u32 ec[CAPD];
    __shared__ u64 wt64[NWAVE];
    __shared__ double wtf[NWAVE];
    __shared__ u32 dcount, ecnt, over;
    const int tid = threadIdx.x;
    const u32 b = blockIdx.x, P = w.P;
    const BinMap bm = bin_map(w.ctl, P);
    for (u32 i = tid; i < (u32)BSLOTS; i += WG) {
        tk[i] = EMPTY;
        tc[i] = 0;
    }
    if (tid == 0) dcount = ecnt = over = 0;
    __syncthreads();
    auto insert = [&](u64 key, u32 c) {
        u32 s = (mix32(key) >> 8) & (BSLOTS - 1);
        for (u32 probe = 0; probe < (u32)BSLOTS; ++probe) {
            if (__hip_atomic_load(&over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
            const u64 cur = __hip_atomic_load(&tk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == key) {
                atomicAdd(&tc[s], c);
                return;
            }
            if (cur == EMPTY) {
                const u64 old = atomicCAS(&tk[s], EMPTY, key);
                if (old == EMPTY) {
                    if (atomicAdd(&dcount, 1u) >= (u32)CAPD) atomicOr(&over, 1u);
                    atomicAdd(&tc[s], c);
                    return;
                }
                if (old == key) {
                    atomicAdd(&tc[s], c);
                    return;
                }
            }
            s = (s + 1) & (BSLOTS - 1);
        }
        atomicOr(&over, 1u);
    };
    // the heavy values that fall into this bin, with the copies k_rl_strip counted
    const u32 H = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
    for (u32 i = tid; i < H; i += WG) {
        const u64 v = w.heavy[i];
        const u32 c = w.hcnt[i];
        if (c && v != EMPTY && bin_of(bm, v) == b) insert(v, c);
    }
    // the bin's segment of every batch
    for (u32 r = tid; r < w.regions; r += WG) {
        const u32 nbt = w.nbatch[r] <= w.nb ? w.nbatch[r] : 0u;
        for (u32 j = 0; j < nbt; ++j) {
            const u32 *row = w.start + ((size_t)r * w.nb + j) * (P + 1);
            const u32 s0 = row[b], s1 = row[b + 1];
            const u64 base = (u64)r * w.region + (u64)j * TBUF;
            if (s1 > (u32)TBUF || s0 > s1 || base + s1 > (u64)w.n) {  // (offsets are checked before they are believed)
                atomicOr(&w.ctl->err, ERR_RANGE);
                continue;
            }
            for (u32 i = s0; i < s1; ++i) insert(w.tail[base + i], 1u);
        }
    }
    __syncthreads();
    // (one thread reads the error word for all: another workgroup may be setting it at this moment)
    if (__syncthreads_or(tid == 0 && (over || w.ctl->err))) {  // more distinct values than a bin holds, or an earlier failure: no index
        if (tid == 0) {
            if (over) atomicOr(&w.ctl->err, ERR_BIN);
            w.bin_d[b] = 0;
            w.bin_c[b] = 0;
            w.bin_w[b] = 0.0;
        }
        return;
    }
    for (u32 s = tid; s < (u32)BSLOTS; s += WG) {
        const u64 v = tk[s];
        if (v != EMPTY) {
            const u32 p = atomicAdd(&ecnt, 1u);
            if (p < (u32)CAPD) {
                ev[p] = v;
                ec[p] = tc[s];
            }
        }
    }
    __syncthreads();
    const u32 d = ecnt < (u32)CAPD ? ecnt : (u32)CAPD;
    // the values are distinct: rank = number of smaller ones (a bin holds a few hundred values as a rule)
    for (u32 i = tid; i < d; i += WG) {
        const u64 v = ev[i];
        u32 r = 0;
#pragma unroll 4
        for (u32 j = 0; j < d; ++j) r += ev[j] < v ? 1u : 0u;
        tk[r] = v;
        tc[r] = ec[i];
    }
    __syncthreads();
    // running counts and weights in ascending order: every thread takes `per` consecutive runs
    const u32 per = (d + WG - 1) / WG;  // <= CAPD / WG = 4
    const u32 i0 = tid * per < d ? tid * per : d, i1 = i0 + per < d ? i0 + per : d;
    u64 cs = 0;
    double ws = 0.0;
    for (u32 i = i0; i < i1; ++i) {
        cs += tc[i];
        ws += (double)tc[i] * exp2(__longlong_as_double((long long)tk[i]) - w.logn);
    }
    u64 ctot;
    double wtotal;
    u64 cex = wg_excl_scan_t<u64>(cs, wt64, ctot);
    double wex = wg_excl_scan_t<double>(ws, wtf, wtotal);
    u64 *rv = w.rval + (size_t)b * CAPD;
    u64 *rc = w.rccum + (size_t)b * CAPD;
    double *rw = w.rwcum + (size_t)b * CAPD;
    for (u32 i = i0; i < i1; ++i) {
        cex += tc[i];
        wex += (double)tc[i] * exp2(__longlong_as_double((long long)tk[i]) - w.logn);
        rv[i] = tk[i];
        rc[i] = cex;
        rw[i] = wex;
    }
    if (tid == 0) {
        w.bin_d[b] = d;
        w.bin_c[b] = ctot;
        w.bin_w[b] = wtotal;
    }
}

// ---- prefix over the bins, runs to their final places ---------------------------------------------------------
__global__ void __launch_bounds__(WG) k_rl_final(RWs w) {
    __shared__ u32 wt32[NWAVE];
    __shared__ u64 wt64[NWAVE];
    __shared__ double wtf[NWAVE];
    __shared__ u32 bd[P_MAX];
    __shared__ u64 bc[P_MAX];
    __shared__ double bw[P_MAX];
    const int tid = threadIdx.x;
    const u32 b = blockIdx.x, P = w.P;
    __shared__ u32 s_err;
    if (tid == 0) s_err = w.ctl->err;  // (as the kernel found it: workgroup 0 may add to it below)
    __syncthreads();
    const u32 err = s_err;
    // every workgroup scans the same values with the same additions: the bases agree
    const u32 j0 = 2 * tid, j1 = 2 * tid + 1;
    const u32 d0 = j0 < P ? w.bin_d[j0] : 0u, d1 = j1 < P ? w.bin_d[j1] : 0u;
    const u64 c0 = j0 < P ? w.bin_c[j0] : 0ull, c1 = j1 < P ? w.bin_c[j1] : 0ull;
    const double w0 = j0 < P ? w.bin_w[j0] : 0.0, w1 = j1 < P ? w.bin_w[j1] : 0.0;
    u32 dtot;
    u64 ctot;
    double wtotal;
    const u32 dex = wg_excl_scan(d0 + d1, wt32, dtot);
    const u64 cex = wg_excl_scan_t<u64>(c0 + c1, wt64, ctot);
    const double wex = wg_excl_scan_t<double>(w0 + w1, wtf, wtotal);
    bd[j0] = dex;
    bd[j1] = dex + d0;
    bc[j0] = cex;
    bc[j1] = cex + c0;
    bw[j0] = wex;
    bw[j1] = wex + w0;
    __syncthreads();
    const u32 allones = w.ctl->n_allones;
    if (b == 0 && tid == 0) {
        u32 e = err;
        if (!e && ctot + allones != (u64)w.n) e = ERR_COUNT;  // (keys changed under the build)
        if (!e && (u64)dtot + (allones ? 1u : 0u) > w.out_cap) e = ERR_RANGE;
        if (e && !err) atomicOr(&w.ctl->err, e);
        w.out_info[0] = e ? 0ull : (u64)dtot + (allones ? 1u : 0u);
        w.out_info[1] = e;
        w.out_info[2] = ctot + allones;
        w.out_info[3] = w.ctl->n_heavy;
        w.out_info[4] = w.ctl->n_tail;
        w.out_info[5] = w.ctl->list_age;
        w.out_info[6] = 0;
        w.out_info[7] = 0;
    }
    if (err) return;
    const u32 d = w.bin_d[b] <= (u32)CAPD ? w.bin_d[b] : 0u;
    const u32 dbase = bd[b];
    const u64 cbase = bc[b];
    const double wbase = bw[b];
    const u64 *rv = w.rval + (size_t)b * CAPD;
    const u64 *rc = w.rccum + (size_t)b * CAPD;
    const double *rw = w.rwcum + (size_t)b * CAPD;
    for (u32 i = tid; i < d; i += WG) {
        const u64 at = (u64)dbase + i;
        if (at < w.out_cap) {
            w.out_vals[at] = rv[i];
            w.out_ccum[at] = cbase + rc[i];
            w.out_wcum[at] = wbase + rw[i];
        }
    }
    if (b == P - 1 && allones && tid == 0 && (u64)dtot < w.out_cap) {  // the keys that are all ones: the last run
        w.out_vals[dtot] = EMPTY;
        w.out_ccum[dtot] = ctot + allones;
        w.out_wcum[dtot] = __longlong_as_double(0x7FF8000000000000ll);  // (as a double the key is a NaN: so is its weight)
    }
}

// ---- the sorted keys and their positions, for callers that ask for them ----------------------------------------
// keys_out[i] = the i-th smallest key, pos_out[i] = the running sum up to and including it: inside a run of c copies
// of v that starts at rank s, after cum_pos of the run before, (i - s + 1) * 2^(v - log2 n) more.
__global__ void __launch_bounds__(256) k_rl_expand(const u64 *__restrict__ vals, const u64 *__restrict__ ccum,
                                                    const double *__restrict__ wcum, const u64 *__restrict__ info, u64 n,
                                                    double logn, u64 *__restrict__ keys_out, double *__restrict__ pos_out) {
    constexpr u32 CH = 2048;
    __shared__ u64 sc[CH + 2];
    __shared__ u64 s_d0, s_d1;
    const u64 D = info[0];
    const bool bad = info[1] != 0 || D == 0;
    const u64 nan = 0x7FF8000000000000ull;
    for (u64 i0 = (u64)blockIdx.x * CH; i0 < n; i0 += (u64)gridDim.x * CH) {
        const u64 i1 = i0 + CH < n ? i0 + CH : n;
        if (bad) {  // a failed build must not pass for a result
            for (u64 i = i0 + threadIdx.x; i < i1; i += 256) {
                if (keys_out) keys_out[i] = nan;
                if (pos_out) pos_out[i] = __longlong_as_double((long long)nan);
            }
            continue;
        }
        __syncthreads();
        if (threadIdx.x == 0 || threadIdx.x == 64) {  // first run whose cumulative count exceeds the rank
            const u64 x = threadIdx.x == 0 ? i0 : i1 - 1;
            u64 lo = 0, hi = D;
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (ccum[mid] > x) hi = mid; else lo = mid + 1;
            }
            if (lo >= D) lo = D - 1;  // (counts that do not reach n: k_rl_final has reported it)
            if (threadIdx.x == 0) s_d0 = lo; else s_d1 = lo;
        }
        __syncthreads();
        const u64 d0 = s_d0, d1 = s_d1;
        const u32 nr = (u32)(d1 - d0 + 1);  // <= CH: every run holds a key of the chunk
        for (u32 j = threadIdx.x; j <= nr && j <= CH; j += 256) sc[j] = d0 + j ? ccum[d0 + j - 1] : 0ull;
        __syncthreads();
        for (u64 i = i0 + threadIdx.x; i < i1; i += 256) {
            u32 lo = 1, hi = nr < CH ? nr : CH;  // first j in [1, nr] with sc[j] > i
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (sc[mid] > i) hi = mid; else lo = mid + 1;
            }
            const u64 dd = d0 + lo - 1;
            const u64 v = vals[dd];
            if (keys_out) keys_out[i] = v;
            if (pos_out) {
                const double before = dd ? wcum[dd - 1] : 0.0;
                pos_out[i] = before + (double)(i - sc[lo - 1] + 1) * exp2(__longlong_as_double((long long)v) - logn);
            }
        }
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct RLayout {
    size_t ctl, heavy, hcnt, nbatch, bin_d, bin_c, bin_w, start, rval, rccum, rwcum, tail, end;
    u32 region, regions, nb, P;
};

RLayout layout_for(u64 n) {
    RLayout L;
    // keys per strip region: whole tiles, as few as give every k_rl_strip workgroup the GPU holds at once one region
    const u64 tiles = (n + TILE - 1) / TILE;
    const u64 per = (tiles + STRIP_SLOTS - 1) / STRIP_SLOTS;
    L.region = (u32)((per ? per : 1) * TILE);
    L.regions = (u32)((n + L.region - 1) / L.region);
    if (L.regions == 0) L.regions = 1;
    L.nb = L.region / TBUF;
    // bins: a bin costs a workgroup of k_rl_bins whatever it holds, and a bin may hold CAPD distinct values
    L.P = n >= (1ull << 21) ? 1024u : n >= (1ull << 17) ? 256u : 64u;
    size_t p = 0;
    L.ctl = p; p += align256(sizeof(RCtl));
    L.heavy = p; p += align256(H_MAX * sizeof(u64));
    L.hcnt = p; p += align256(H_MAX * sizeof(u32));
    L.nbatch = p; p += align256((size_t)L.regions * sizeof(u32));
    L.bin_d = p; p += align256(P_MAX * sizeof(u32));
    L.bin_c = p; p += align256(P_MAX * sizeof(u64));
    L.bin_w = p; p += align256(P_MAX * sizeof(double));
    L.start = p; p += align256((size_t)L.regions * L.nb * (L.P + 1) * sizeof(u32));
    L.rval = p; p += align256((size_t)L.P * CAPD * sizeof(u64));
    L.rccum = p; p += align256((size_t)L.P * CAPD * sizeof(u64));
    L.rwcum = p; p += align256((size_t)L.P * CAPD * sizeof(double));
    L.tail = p; p += align256((size_t)L.regions * L.region * sizeof(u64));
    L.end = p;
    return L;
}

}  // namespace

extern "C" size_t mc_rank_runs_workspace_bytes(uint64_t n) { return n && n <= RUNS_MAX_N ? layout_for(n).end : 0; }

extern "C" uint64_t mc_rank_runs_max_keys(void) { return RUNS_MAX_N; }

extern "C" uint64_t mc_rank_runs_capacity(uint64_t n) {
    if (n == 0 || n > RUNS_MAX_N) return 0;
    const u64 most = (u64)layout_for(n).P * CAPD;
    return (n < most ? n : most) + 1;
}

extern "C" int mc_rank_runs_build(const double *keys_in, uint64_t n, uint64_t n_global, double *vals,
                                  uint64_t *cum_count, double *cum_pos, uint64_t capacity, uint64_t *info,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (!info) {
        mc_set_error("mc_rank_runs_build: info is null");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {  // an index of no runs
        HIP_TRY(hipMemsetAsync(info, 0, 8 * sizeof(uint64_t), s));
        return MC_OK;
    }
    if (!keys_in || !vals || !cum_count || !cum_pos || !workspace || n_global < n) {
        mc_set_error("mc_rank_runs_build: bad argument");
        return MC_ERR_ARG;
    }
    if (n > RUNS_MAX_N) {
        mc_set_error("mc_rank_runs_build: %llu keys, the run index takes at most %llu (use mc_rank_build beyond that)",
                     (unsigned long long)n, (unsigned long long)RUNS_MAX_N);
        return MC_ERR_ARG;
    }
    const RLayout L = layout_for(n);
    if (workspace_bytes < L.end) {
        mc_set_error("mc_rank_runs_build: workspace of %zu bytes, need %zu", workspace_bytes, L.end);
        return MC_ERR_WORKSPACE;
    }
    if (capacity < mc_rank_runs_capacity(n)) {
        mc_set_error("mc_rank_runs_build: output arrays of %llu entries, need %llu (mc_rank_runs_capacity)",
                     (unsigned long long)capacity, (unsigned long long)mc_rank_runs_capacity(n));
        return MC_ERR_ARG;
    }
    char *base = (char *)workspace;
    RWs w;
    w.ctl = (RCtl *)(base + L.ctl);
    w.heavy = (u64 *)(base + L.heavy);
    w.hcnt = (u32 *)(base + L.hcnt);
    w.in = (const u64 *)keys_in;
    w.tail = (u64 *)(base + L.tail);
    w.nbatch = (u32 *)(base + L.nbatch);
    w.start = (u32 *)(base + L.start);
    w.rval = (u64 *)(base + L.rval);
    w.rccum = (u64 *)(base + L.rccum);
    w.rwcum = (double *)(base + L.rwcum);
    w.bin_d = (u32 *)(base + L.bin_d);
    w.bin_c = (u64 *)(base + L.bin_c);
    w.bin_w = (double *)(base + L.bin_w);
    w.n = (u32)n;
    w.region = L.region;
    w.regions = L.regions;
    w.nb = L.nb;
    w.P = L.P;
    w.logn = std::log2((double)n_global);
    static const int env_reuse = getenv("MC_SORT_REUSE") ? atoi(getenv("MC_SORT_REUSE")) : 1;
    w.reuse_list = env_reuse ? 1u : 0u;
    w.debug_fail = 0;
    w.out_vals = (u64 *)vals;
    w.out_ccum = (u64 *)cum_count;
    w.out_wcum = cum_pos;
    w.out_info = (u64 *)info;
    w.out_cap = capacity;
    hipLaunchKernelGGL(k_rl_sniff, dim3(1), dim3(SNIFF_THREADS), 0, s, w);
    hipLaunchKernelGGL(k_rl_strip, dim3(L.regions < (u32)STRIP_SLOTS ? L.regions : (u32)STRIP_SLOTS), dim3(WG), 0, s, w);
    hipLaunchKernelGGL(k_rl_bins, dim3(L.P), dim3(WG), 0, s, w);
    hipLaunchKernelGGL(k_rl_final, dim3(L.P), dim3(WG), 0, s, w);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_rank_runs_forget(void *workspace, uint64_t n, void *stream) {
    if (!workspace || n == 0 || n > RUNS_MAX_N) {
        mc_set_error("mc_rank_runs_forget: null workspace or a key count outside 1 .. 2^26");
        return MC_ERR_ARG;
    }
    HIP_TRY(hipMemsetAsync((char *)workspace + layout_for(n).ctl, 0, sizeof(RCtl), (hipStream_t)stream));
    return MC_OK;
}

extern "C" int mc_rank_runs_check(const uint64_t *info, void *stream, uint64_t host_info[8]) {
    uint64_t local[8] = {0};
    if (!info) {
        mc_set_error("mc_rank_runs_check: info is null");
        return MC_ERR_ARG;
    }
    HIP_TRY(hipMemcpyAsync(local, info, sizeof local, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (host_info) for (int i = 0; i < 8; ++i) host_info[i] = local[i];
    if (local[1] & ERR_BIN) {
        mc_set_error("mc_rank_runs_build: these keys do not suit the run index (a bin of the value range holds more "
                     "than %d distinct values); queries answer NaN -- build with mc_rank_build instead", CAPD);
        return MC_ERR_UNSUITED;
    }
    if (local[1]) {
        mc_set_error("mc_rank_runs_build failed on the device (error word 0x%llx: an offset or a count did not match "
                     "the keys -- were they overwritten while the build ran?); queries answer NaN",
                     (unsigned long long)local[1]);
        return MC_ERR_WORKSPACE;
    }
    return MC_OK;
}

extern "C" int mc_rank_runs_expand(const double *vals, const uint64_t *cum_count, const double *cum_pos,
                                   const uint64_t *info, uint64_t n, uint64_t n_global, double *keys_sorted,
                                   double *positions, void *stream) {
    if (n == 0) return MC_OK;
    if (!vals || !cum_count || !cum_pos || !info || n_global < n || (!keys_sorted && !positions)) {
        mc_set_error("mc_rank_runs_expand: bad argument");
        return MC_ERR_ARG;
    }
    const u64 chunks = (n + 2047) / 2048;
    hipLaunchKernelGGL(k_rl_expand, dim3((u32)(chunks < 4096 ? chunks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const u64 *)vals, (const u64 *)cum_count, cum_pos, (const u64 *)info, (u64)n,
                       std::log2((double)n_global), (u64 *)keys_sorted, positions);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}
