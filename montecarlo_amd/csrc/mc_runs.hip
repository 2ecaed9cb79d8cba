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
//                correct index; stale ones only cost time).  The heavy values are also laid out as a two-choice
//                hash table (each in one of its two slots) that k_rl_strip copies into LDS.
//   k_rl_strip   ONE read of all keys.  A copy of a heavy value is counted in LDS (two reads of the table, no walk).  Every other key ("tail") goes to
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
#include <cstring>

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
constexpr int CELLS = 4096;        // equal cells of the value range; a bin is a run of cells (the map's look-up table)
constexpr int CAPD = 2048;         // distinct values a bin may hold (k_rl_bins: MAIN_CAP + OVCAP)
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
    u32 bad_bin, bad_main, bad_over, bad_pad;  // a bin that refused its keys: which, its occupied sub-ranges, its overflow values
};

struct RWs {
    RCtl *ctl;
    u64 *heavy;      // [H_MAX] ascending
    u32 *hcnt;       // [H_MAX] their copies
    u64 *htab;       // [HSET] the heavy values as a two-choice table (k_rl_sniff builds it, k_rl_strip copies it to LDS)
    u32 *hidx;       // [HSET] slot -> index into `heavy`
    const u64 *in;
    u64 *tail;       // [n] region r's batches at r * region + j * TBUF, each ordered by bin
    u32 *nbatch;     // [regions] batches region r wrote
    u64 *rmax;       // [regions] largest finite tail key of region r (bit pattern; 0: none)
    u32 *hbin;       // [P + 1] first heavy value (index into `heavy`) that lies in bin b or above
    unsigned short *lut;    // [CELLS] bin of every cell of the value range (monotone)
    unsigned short *cell0;  // [P + 1] first cell of bin b (cell0[P] = CELLS; an empty bin: cell0[b] = cell0[b + 1])
    u32 *start;      // [regions * nb][P + 1] offsets of the bins inside a batch
    u64 *rval;       // [P * CAPD] a bin's distinct values, ascending
    u64 *rccum;      // [P * CAPD] ... copies up to and including each, inside the bin
    double *rwcum;   // [P * CAPD] ... and their weights
    u32 *bin_d;      // [P_MAX] distinct values of the bin
    u64 *bin_c;      // [P_MAX] keys of the bin
    double *bin_w;   // [P_MAX] their weights
    u32 n, region, regions, nb, P;
    double logn;
    // the look-up table of k_rl_sniff: a cell's bin follows, in these shares, the sample KEYS below it, the sample's
    // DISTINCT values below it, and (the rest) its own place in the range.  0.25 / 0 for a build of up to 2^26 keys
    double eq_share, eq_distinct;
    u32 reuse_list, debug_fail;
    // outputs (the caller's)
    u64 *out_vals, *out_ccum;
    double *out_wcum;
    u64 *phase;      // development aid (mc_rank_runs_debug_phases): per-phase clock sums of k_rl_bins / k_rl_strip, or null
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
// the two places a heavy value may have in the strip's table (two-choice hashing: a look-up is two reads, never a walk)
__device__ __forceinline__ void strip_slots(u64 k, u32 &s1, u32 &s2) {
    const u32 x = mix32(k);
    s1 = x >> 21;
    s2 = (x >> 10) & (u32)(HSET - 1);
}

// The bin of a key: the range [lo, hi] the sample saw (widened) is cut into CELLS equal cells, and a look-up table
// made from the sample's own distribution gives every cell its bin, so that the bins hold about equally many keys
// (equal-width bins held up to seven times the mean where the scores are densest, and the workgroups of those bins
// set k_rl_bins' time).  Monotone in the key's BIT pattern (the order the index is kept in; for the non-negative
// doubles it is made for, the numeric order): below lo -> cell 0, at or above hi (infinities, NaNs, a set sign bit)
// -> the last cell.
struct BinMap {
    u64 lo_bits, hi_bits;
    double lo, cscale;
    const unsigned short *lut;  // (LDS or memory)
};
__device__ __forceinline__ BinMap bin_map(double lo, double hi, const unsigned short *lut) {
    BinMap m;
    m.lo = lo;
    m.lo_bits = (u64)__double_as_longlong(m.lo);
    m.hi_bits = (u64)__double_as_longlong(hi);
    m.cscale = hi > m.lo ? (double)CELLS / (hi - m.lo) : 0.0;
    m.lut = lut;
    return m;
}
__device__ __forceinline__ u32 cell_of(const BinMap &m, u64 kb) {
    if (kb >= m.hi_bits) return CELLS - 1;
    if (kb <= m.lo_bits) return 0u;
    const double x = (__longlong_as_double((long long)kb) - m.lo) * m.cscale;
    const u32 c = (u32)x;
    return c < (u32)CELLS ? c : (u32)CELLS - 1;
}
__device__ __forceinline__ u32 bin_of(const BinMap &m, u64 kb) { return m.lut[cell_of(m, kb)]; }

// ---- the sample: heavy values and the range of the rest ------------------------------------------------------
__device__ __forceinline__ bool sample_index(u32 n, u32 q, u64 &idx) {
    if (n <= (u32)SNIFF_S) {
        idx = q;
        return q < n;
    }
    idx = (u64)(q / SNIFF_BLOCK) * (n / (SNIFF_S / SNIFF_BLOCK)) + (q % SNIFF_BLOCK);  // (blocks of 16: 1024 pages)
    return true;
}

__device__ __forceinline__ void rl_sniff_body(const RWs &w) {
    constexpr int PER = SNIFF_S / SNIFF_THREADS;
    static_assert(SNIFF_SLOTS == 16384, "slot = top 14 bits of the mix");
    __shared__ u32 tag[SNIFF_SLOTS];     // 0 = empty
    __shared__ u32 tc[SNIFF_SLOTS / 2];  // two 16-bit counts per word
    __shared__ u32 claim[SNIFF_SLOTS / 32];
    __shared__ u64 list[H_MAX];
    __shared__ u64 red[2 * (SNIFF_THREADS / 64)];
    __shared__ u32 nlist;
    __shared__ u32 keep_h, keep_tail, keep_age;
    __shared__ u64 keep_lo, keep_hi;
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
        keep_lo = (u64)__double_as_longlong(c.map_lo);
        keep_hi = (u64)__double_as_longlong(c.map_hi);
    }
    __syncthreads();
    u32 keep = keep_h;
    if (keep != 0xFFFFFFFFu) {
        // a look at a thousand keys: does the range the hint describes still hold them?  (Keys of another source far
        // outside it would all fall into the first or the last bin, more than a bin takes: the build would refuse
        // keys it takes after a search.)
        const u64 kb = w.in[(u64)tid * (n / SNIFF_THREADS) + (n < (u32)SNIFF_THREADS ? (u32)tid % n : 0u)];
        const int out = kb < F64_INF && (kb < keep_lo || kb >= keep_hi);
        if (__syncthreads_count(out) * 4 > SNIFF_THREADS) keep = 0xFFFFFFFFu;
    }
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
    // the range of the sample's finite keys, heavy ones included (the next build asks of its keys whether this
    // range still holds them, and most keys are copies of heavy values)
    {
        u64 mn = EMPTY, mx = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (have[q] && k[q] < F64_INF) {
                mn = k[q] < mn ? k[q] : mn;
                mx = k[q] > mx ? k[q] : mx;
            }
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) {
            red[wid] = mn;
            red[SNIFF_THREADS / 64 + wid] = mx;
        }
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
        for (u32 i2 = tid; i2 < H; i2 += SNIFF_THREADS) {
            w.heavy[rank[i2]] = list[i2];
            ((u64 *)tag)[rank[i2]] = list[i2];  // (the tags are not needed any more: the list in ascending order)
        }
    }
    __shared__ double s_lo, s_hi;
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
            // full input reaches, spread over cells of their own instead of piling up in the first and the last
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
        s_lo = lo;
        s_hi = hi;
    }
    // ---- the look-up table: cells of the range -> bins of about equal numbers of (non-heavy) sample keys ----
    static_assert(CELLS == 4 * SNIFF_THREADS, "a thread scans four cells");
    static_assert(SNIFF_SLOTS / 2 >= CELLS && SNIFF_SLOTS >= 2 * H_MAX + CELLS / 2, "the histogram borrows tc, the table borrows tag");
    u32 *hist = tc;                                        // (the ranks that lived here are dead after the barrier below)
    unsigned short *slut = (unsigned short *)(tag + 2 * H_MAX);  // (behind the sorted list)
    __shared__ u32 wtot16[SNIFF_THREADS / 64], wtotd[SNIFF_THREADS / 64];
    // the same histogram in DISTINCT VALUES, estimated: a key is -log2 p of a sampled string, and a string of that
    // probability is drawn n * 2^-key times in n draws -- a key above log2 n stands for a value of its own, a key
    // below it for 2^(key - log2 n) of one (16-bit fixed point).  Keys that are no such scores only make the bins
    // uneven: a bin that outgrows its table refuses the build.
    __shared__ u32 histd[CELLS];
    __syncthreads();
    for (u32 i = tid; i < (u32)CELLS; i += SNIFF_THREADS) {
        hist[i] = 0;
        histd[i] = 0;
    }
    __syncthreads();
    {
        const BinMap cm = bin_map(s_lo, s_hi, nullptr);
        const double ln = log2((double)n);
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (have[q] && k[q] < F64_INF && !over[q]) {
                const u32 cell = cell_of(cm, k[q]);
                atomicAdd(&hist[cell], 1u);
                if (w.eq_distinct > 0.0) {
                    const double v = __longlong_as_double((long long)k[q]);
                    const double share = v >= ln ? 1.0 : exp2(v - ln);
                    atomicAdd(&histd[cell], (u32)(share * 65536.0 + 0.5));
                }
            }
    }
    __syncthreads();
    {
        const u32 c4[4] = {hist[4 * tid], hist[4 * tid + 1], hist[4 * tid + 2], hist[4 * tid + 3]};
        const u32 d4[4] = {histd[4 * tid], histd[4 * tid + 1], histd[4 * tid + 2], histd[4 * tid + 3]};
        const u32 mine = c4[0] + c4[1] + c4[2] + c4[3], mined = d4[0] + d4[1] + d4[2] + d4[3];
        u32 inc = mine, incd = mined;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(inc, o), td = __shfl_up(incd, o);
            if (lane >= o) {
                inc += t;
                incd += td;
            }
        }
        if (lane == 63) {
            wtot16[wid] = inc;
            wtotd[wid] = incd;
        }
        __syncthreads();
        u32 off = 0, total = 0, offd = 0, totald = 0;
        for (int i = 0; i < SNIFF_THREADS / 64; ++i) {
            if (i < wid) {
                off += wtot16[i];
                offd += wtotd[i];
            }
            total += wtot16[i];
            totald += wtotd[i];
        }
        u32 before = off + inc - mine;  // sample keys in the cells before this thread's four
        u32 befored = offd + incd - mined;  // ... and distinct sample values
        // a cell's bin: a quarter from the share of the sample below it, three quarters from the cell's own place in
        // the range.  Equal-width bins hold up to 43,000 of the headline step's 1.96e6 tail keys (the mean is 1900)
        // and at most 921 distinct values; bins of equal numbers of keys grow wide where keys are few but all
        // different (3127 distinct values in one); at a quarter the fullest bin holds 19,000 keys and 1231 values
        // (scripts/bin_stats.py)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 c = 4 * tid + j;
            const double lin = (double)c / (double)CELLS;
            double f = total ? w.eq_share * ((double)before / (double)total) + (1.0 - w.eq_share) * lin : lin;
            if (w.eq_distinct > 0.0 && total && totald)
                f = w.eq_share * ((double)before / (double)total) + w.eq_distinct * ((double)befored / (double)totald) +
                    (1.0 - w.eq_share - w.eq_distinct) * lin;
            u32 bin = (u32)(f * (double)w.P);
            if (bin >= w.P) bin = w.P - 1;
            slut[c] = (unsigned short)bin;
            w.lut[c] = (unsigned short)bin;
            before += c4[j];
            befored += d4[j];
        }
    }
    __syncthreads();
    {   // where every bin starts among the cells and among the heavy values (the map is monotone: both are runs)
        const BinMap bm = bin_map(s_lo, s_hi, slut);
        const u64 *sorted = (const u64 *)tag;
        for (u32 b = tid; b <= w.P; b += SNIFF_THREADS) {
            u32 lo = 0, hi = CELLS;  // first cell whose bin is >= b
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (slut[mid] >= b) hi = mid; else lo = mid + 1;
            }
            w.cell0[b] = (unsigned short)(b == w.P ? (u32)CELLS : lo);
            lo = 0, hi = H;  // first i with bin_of(sorted[i]) >= b
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (bin_of(bm, sorted[mid]) >= b) hi = mid; else lo = mid + 1;
            }
            w.hbin[b] = b == w.P ? H : lo;
        }
    }
    // ---- the heavy values as a two-choice table for k_rl_strip: every value sits in one of its two slots, so a key
    //      is looked up with two reads and no walk (a linear-probing table at a fifth full made four keys in a
    //      hundred walk, which is a branch taken by nearly every wave).  Values take a free slot of theirs at once;
    //      the few that find both taken are placed one after the other, moving others to their second slots.  A
    //      value that finds no place after 64 moves is left out: its copies then count as tail keys, which is slower
    //      and as correct (hcnt stays 0 and k_rl_bins skips it).
    __syncthreads();
    if (w.htab) {  // (uniform; the first level of a large build keeps its own table: k_big_strip)
        static_assert(SNIFF_SLOTS / 2 >= 2 * HSET + HSET, "the table borrows tc");
        u64 *T = (u64 *)tc;                 // [HSET]   (the histogram that lived here is dead)
        u32 *I = tc + 2 * HSET;             // [HSET]
        __shared__ u32 late[64];
        __shared__ u32 nlate;
        const u64 *sorted = (const u64 *)tag;
        for (u32 i = tid; i < (u32)HSET; i += SNIFF_THREADS) {
            T[i] = EMPTY;
            I[i] = 0;
        }
        if (tid == 0) nlate = 0;
        __syncthreads();
        for (u32 i = tid; i < H; i += SNIFF_THREADS) {
            const u64 v = sorted[i];
            if (v == EMPTY) continue;
            u32 s1, s2;
            strip_slots(v, s1, s2);
            const u64 o1 = atomicCAS(&T[s1], EMPTY, v);
            if (o1 == EMPTY) {
                I[s1] = i;
            } else if (o1 != v) {
                const u64 o2 = atomicCAS(&T[s2], EMPTY, v);
                if (o2 == EMPTY) {
                    I[s2] = i;
                } else if (o2 != v) {
                    const u32 at = atomicAdd(&nlate, 1u);
                    if (at < 64u) late[at] = i;  // (beyond 64: left out)
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            const u32 nl = nlate < 64u ? nlate : 64u;
            for (u32 j = 0; j < nl; ++j) {
                u64 cur = sorted[late[j]];
                u32 curi = late[j];
                u32 s1, s2;
                strip_slots(cur, s1, s2);
                u32 pos = s1;
                for (int step = 0; step < 64; ++step) {
                    const u64 ev = T[pos];
                    const u32 evi = I[pos];
                    T[pos] = cur;
                    I[pos] = curi;
                    if (ev == EMPTY) break;
                    cur = ev;  // the value that lay there moves to its other slot (left out if the moves run out)
                    curi = evi;
                    strip_slots(cur, s1, s2);
                    pos = pos == s1 ? s2 : s1;
                }
            }
        }
        __syncthreads();
        for (u32 i = tid; i < (u32)HSET; i += SNIFF_THREADS) {
            w.htab[i] = T[i];
            w.hidx[i] = I[i];
        }
    }
}

// ---- strip: count the copies of heavy values, order the rest by bin ------------------------------------------
__device__ __forceinline__ void rl_strip_body(const RWs &w) {
    static_assert(HSET == 2048, "strip_hash yields 11 bits");
    static_assert(TBUF == TILE && TBUF == ITEMS * WG, "a batch is flushed by the whole workgroup, ITEMS keys a lane");
    static_assert(P_MAX == 2 * WG, "the scan of the bins' counts takes two bins a thread");
    __shared__ u64 hk[HSET];       // the heavy values, each in one of its two slots
    __shared__ u32 hs[HSET];       // copies counted per slot
    __shared__ u32 hi[HSET];       // slot -> index of the heavy value
    __shared__ u64 tb[TBUF];       // tail keys waiting for their batch
    __shared__ u32 bh[P_MAX + 1];  // keys per bin of the batch, then the bins' offsets
    __shared__ unsigned short slut[CELLS];  // the map's look-up table
    __shared__ u32 wtot[NWAVE];
    __shared__ u64 wmx[NWAVE];
    __shared__ u32 tfill;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 H = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
    const u32 n = w.n, P = w.P;
    const BinMap bm = bin_map(w.ctl->map_lo, w.ctl->map_hi, slut);
    // the table of the heavy values as k_rl_sniff laid it out (two-choice hashing: strip_slots)
    for (u32 i = tid; i < (u32)HSET; i += WG) {
        hk[i] = H ? w.htab[i] : EMPTY;
        hi[i] = H ? w.hidx[i] : 0u;
        hs[i] = 0;
    }
    for (u32 i = tid; i < (u32)CELLS / 2; i += WG) ((u32 *)slut)[i] = ((const u32 *)w.lut)[i];
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
        // the keys change places inside the buffer (every lane holds its own in registers) and leave it in order,
        // consecutive lanes on consecutive addresses (written from where they lay, every key was a request of its
        // own to the L2: 1.7e6 of them for the 1.96e6 tail keys of the headline step)
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            if (i < m) {
                const u32 at = bh[bin[q]] + rk[q];
                if (at < m) tb[at] = key[q]; else atomicOr(&w.ctl->err, ERR_RANGE);
            }
        }
        __syncthreads();
        if (m > room) {
            if (tid == 0) atomicOr(&w.ctl->err, ERR_RANGE);
            m = room;
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            if (i < m) out[i] = tb[i];
        }
        __syncthreads();
    };

    for (u32 r = blockIdx.x; r < w.regions; r += gridDim.x) {
        const u64 lo = (u64)r * w.region;
        const u32 rn = (u32)(n - lo < (u64)w.region ? n - lo : (u64)w.region);
        u32 nbat = 0;
        u64 kmx = 0;  // largest finite tail key this thread saw
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
            // heavy or not: a key's two slots are read together; it is heavy if one of them holds it (no walk)
            u64 tmask[ITEMS];
            u32 total = 0;
            {
                u32 s1[ITEMS], s2[ITEMS];
                u64 a[ITEMS], b[ITEMS];
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    strip_slots(k[q], s1[q], s2[q]);
                    a[q] = hk[s1[q]];
                    b[q] = hk[s2[q]];
                }
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const bool valid = c0 + q * WG + tid < rn;
                    const bool real = valid && k[q] != EMPTY;  // (a key of all ones matches the table's empty marker: never heavy)
                    if (valid && !real) ++allones;
                    const bool hit_a = a[q] == k[q];
                    const bool heavy = real && (hit_a || b[q] == k[q]);
                    if (heavy) atomicAdd(&hs[hit_a ? s1[q] : s2[q]], 1u);
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
                if (tail && k[q] < F64_INF && k[q] > kmx) kmx = k[q];
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
        kmx = wave_max(kmx);
        if (lane == 0) wmx[tid >> 6] = kmx;
        __syncthreads();
        if (tid == 0) {
            u64 m2 = 0;
            for (int i = 0; i < NWAVE; ++i) m2 = wmx[i] > m2 ? wmx[i] : m2;
            w.rmax[r] = m2;
            w.nbatch[r] = nbat < w.nb ? nbat : w.nb;
        }
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

// development aid: where a workgroup's time goes (sums over the workgroups, in ticks of the 100 MHz wall clock)
#define PHASE_MARK(slot)                                                     \
    do {                                                                     \
        if (w.phase && threadIdx.x == 0) {                                   \
            const u64 now_ = wall_clock64();                                 \
            atomicAdd(&w.phase[slot], now_ - t_phase);                       \
            t_phase = now_;                                                  \
        }                                                                    \
    } while (0)

// ---- one bin: count, order, scan ------------------------------------------------------------------------------
// The bin's table is addressed by VALUE: the bin's range is cut into SUB equal sub-ranges and a value's slot is its
// sub-range -- monotone, so that the occupied slots in slot order ARE the distinct values in ascending order: no
// sort.  A bin holds a few hundred distinct values as a rule; two of them in one sub-range (d^2 / 2 SUB pairs: a
// handful) send the later comer to a small hash table (the "overflow"), whose values are ranked by counting and
// merged into the order.  More than OVCAP overflow values or MAIN_CAP occupied sub-ranges -- a thousand distinct
// values in the bin, or values that differ only far below the bin's width -- and the keys do not suit the index
// (ERR_BIN).  Bin 0 reaches down to 0.0 and bin P - 1 up to the largest finite key; infinities, NaNs and negative
// numbers (bit patterns above +inf) have no sub-range and go to the overflow table, which orders by bits.
//
// What decides the time is the length of the chains of dependent memory and LDS operations, not their number (a
// bin is 1900 keys): a lane asks for all the keys of its segment at once and for all their slots at once (one
// compare-and-swap each, the count is added without waiting).  Tried before: a hash table compacted and ranked all
// against all, 512 lanes a bin, the keys found by a search over the segments' prefix sums (74-97 us for the 1024
// bins of 1e7 keys: twenty barriers around chains of LDS round trips); the same with one wave a bin (551 us: every
// round trip exposed).
constexpr int BW = 256;            // lanes of k_rl_bins' workgroup
constexpr int BNW = BW / 64;
constexpr int SUB = 2048;          // sub-ranges of a bin
constexpr int BSL = SUB / BW;      // sub-ranges a lane scans
constexpr int MAIN_CAP = 1536;     // occupied sub-ranges a bin may have
constexpr int OVS = 1024, OVCAP = 512;  // overflow: hash table of OVS slots, at most OVCAP values
constexpr int SEG_MAX = STRIP_SLOTS;   // regions (one segment of a bin each per round of batches)
constexpr int RK = SEG_MAX / BW;   // regions a lane looks after
constexpr int KU = 16;             // key loads of a lane in flight

// inclusive prefix over the workgroup in thread order (every thread calls; wt: BNW entries)
template <typename T>
__device__ __forceinline__ T bins_incl_scan(T v, T *wt, T &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wt[wid] = inc;
    __syncthreads();
    T off = (T)0, tot = (T)0;
#pragma unroll
    for (int i = 0; i < BNW; ++i) {
        const T t = wt[i];
        if (i < wid) off += t;
        tot += t;
    }
    total = tot;
    return off + inc;
}

__device__ __forceinline__ void rl_bins_body(const RWs &w) {
    static_assert(SUB % BW == 0 && OVS % BW == 0 && SEG_MAX % BW == 0, "whole rounds of the workgroup");
    static_assert(MAIN_CAP + OVCAP <= CAPD, "a bin's runs fit its place in the workspace");
    static_assert(SUB + OVCAP < 65536, "entry ids are 16 bits");
    __shared__ u64 tk[SUB];      // value by sub-range
    __shared__ u32 tc[SUB];      // ... its copies
    __shared__ u64 ovk[OVS];     // overflow: hash table while counting, ascending list afterwards
    __shared__ u32 ovc[OVS];
    __shared__ u64 osk[OVCAP];   // (the overflow table, compacted)
    __shared__ u32 osc[OVCAP];
    __shared__ unsigned short ids[MAIN_CAP];          // occupied sub-ranges, ascending
    __shared__ unsigned short ord[MAIN_CAP + OVCAP];  // entry ids in ascending order of value
    __shared__ u32 wt32[BNW];
    __shared__ u64 wt64[BNW];
    __shared__ double wtf[BNW];
    __shared__ u64 wmx[BNW];
    __shared__ u32 nov, over;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 b = blockIdx.x, P = w.P;
    // everything the bin needs from memory is asked for at once: the map, where its heavy values start, and per
    // region its count of batches and its first batch's offsets of bin b (a region holds one batch as a rule)
    const double map_lo = w.ctl->map_lo, map_hi = w.ctl->map_hi;
    const u32 hb0 = w.hbin[b], hb1 = w.hbin[b + 1];
    const u32 cl0 = w.cell0[b], cl1 = w.cell0[b + 1];
    const bool last = cl1 == (u32)CELLS && cl0 < cl1;  // the bin of the last cell: everything from hi upwards is its
    u32 nbt[RK], f0[RK], f1[RK];
    u64 rmx = 0;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const u32 r = k * BW + tid;
        nbt[k] = f0[k] = f1[k] = 0;
        if (r < w.regions) {
            const u32 *row0 = w.start + (size_t)r * w.nb * (P + 1);
            nbt[k] = w.nbatch[r];
            f0[k] = row0[b];
            f1[k] = row0[b + 1];
            if (last) {
                const u64 m = w.rmax[r];
                rmx = m > rmx ? m : rmx;
            }
        }
    }
    u64 t_phase = w.phase ? wall_clock64() : 0ull;
    const u64 t_first = t_phase;
    for (u32 i = tid; i < (u32)SUB; i += BW) {
        tk[i] = EMPTY;
        tc[i] = 0;
    }
    for (u32 i = tid; i < (u32)OVS; i += BW) {
        ovk[i] = EMPTY;
        ovc[i] = 0;
    }
    if (tid == 0) nov = over = 0;
    // the bin's own range [base, top) and its sub-ranges
    const double cw = (map_hi - map_lo) / (double)CELLS;
    const double base = cl0 ? map_lo + (double)cl0 * cw : 0.0;  // (the bin of cell 0 reaches down to 0.0)
    double top = map_lo + (double)cl1 * cw;
    if (last) {  // (uniform) up to the largest finite key there is
        rmx = wave_max(rmx);
        if (lane == 0) wmx[tid >> 6] = rmx;
        __syncthreads();
        u64 m2 = 0;
        for (int i = 0; i < BNW; ++i) m2 = wmx[i] > m2 ? wmx[i] : m2;
        const u32 Hn = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
        for (u32 i = Hn; i > 0 && i + 4 > Hn; --i) {  // (the list's last values: the largest finite one of them)
            const u64 v = w.heavy[i - 1];
            if (v < F64_INF) {
                m2 = v > m2 ? v : m2;
                break;
            }
        }
        const double km = __longlong_as_double((long long)m2);
        if (km >= top) top = km * (1.0 + 0x1.0p-40) + 0x1.0p-1000;  // (strictly above: km itself lands in the last sub-range)
    }
    if (!(top > base)) top = base + 1.0;
    const double sscale = (double)SUB / (top - base);
    auto slot_of = [&](u64 kb) -> u32 {
        if (kb >= F64_INF) return (u32)SUB;  // no sub-range
        const double x = (__longlong_as_double((long long)kb) - base) * sscale;
        if (!(x > 0.0)) return 0u;
        return x < (double)SUB ? (u32)x : (u32)SUB - 1u;
    };
    __syncthreads();
    PHASE_MARK(0);
    auto ov_insert = [&](u64 key, u32 c) {  // (rare)
        u32 h = (mix32(key) >> 9) & (OVS - 1);
        for (u32 probe = 0; probe < (u32)OVS; ++probe) {
            if (__hip_atomic_load(&over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
            const u64 cur = __hip_atomic_load(&ovk[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == key) {
                atomicAdd(&ovc[h], c);
                return;
            }
            if (cur == EMPTY) {
                const u64 old = atomicCAS(&ovk[h], EMPTY, key);
                if (old == EMPTY) {
                    if (atomicAdd(&nov, 1u) >= (u32)OVCAP) atomicOr(&over, 1u);
                    atomicAdd(&ovc[h], c);
                    return;
                }
                if (old == key) {
                    atomicAdd(&ovc[h], c);
                    return;
                }
            }
            h = (h + 1) & (OVS - 1);
        }
        atomicOr(&over, 1u);
    };
    // counts n keys of a lane: all their slots are asked for before the first answer is looked at
    auto count = [&](const u64 (&key)[KU], const u32 (&c)[KU], u32 n) {
        u32 s[KU];
        u64 old[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) s[u] = (u32)u < n ? slot_of(key[u]) : (u32)SUB;
#pragma unroll
        for (int u = 0; u < KU; ++u) old[u] = s[u] < (u32)SUB ? atomicCAS(&tk[s[u]], EMPTY, key[u]) : 0ull;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if ((u32)u >= n) continue;
            if (s[u] < (u32)SUB && (old[u] == EMPTY || old[u] == key[u])) atomicAdd(&tc[s[u]], c[u]);
            else ov_insert(key[u], c[u]);
        }
    };
    // the heavy values of this bin, with the copies k_rl_strip counted (a bin holds none or a few)
    {
        const u32 H = hb1 <= (u32)H_MAX ? hb1 : 0u;
        for (u32 i = hb0 + tid; i < H; i += BW) {
            u64 key[KU] = {w.heavy[i]};
            u32 c[KU] = {w.hcnt[i]};
            if (c[0] && key[0] != EMPTY) count(key, c, 1u);
        }
    }
    PHASE_MARK(1);
    // the bin's segment of every batch: a lane walks the segments of its regions, KU keys at a time
    for (u32 j = 0; j < w.nb; ++j) {
        bool any = false;
        u32 s0[RK], s1[RK];
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const u32 r = k * BW + tid;
            s0[k] = s1[k] = 0;
            if (j < nbt[k] && nbt[k] <= w.nb) {
                any = true;
                if (j == 0) {
                    s0[k] = f0[k];
                    s1[k] = f1[k];
                } else {
                    const u32 *row = w.start + ((size_t)r * w.nb + j) * (P + 1);
                    s0[k] = row[b];
                    s1[k] = row[b + 1];
                }
                const u64 at = (u64)r * w.region + (u64)j * TBUF;
                if (s1[k] > (u32)TBUF || s0[k] > s1[k] || at + s1[k] > (u64)w.n) {  // (offsets are checked before they are believed)
                    atomicOr(&w.ctl->err, ERR_RANGE);
                    s0[k] = s1[k] = 0;
                }
            }
        }
        if (j && !__syncthreads_or(any)) break;  // (uniform; round 0 always runs)
        PHASE_MARK(2);
        u32 one[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) one[u] = 1u;
#pragma unroll
        for (int k = 0; k < RK; ++k) {  // (a segment holds a few keys as a rule, a few dozen where the scores are densest)
            const u64 *src = w.tail + ((u64)(k * BW + tid) * w.region + (u64)j * TBUF);
            for (u32 i0 = s0[k]; i0 < s1[k]; i0 += KU) {
                u64 key[KU];
#pragma unroll
                for (int u = 0; u < KU; ++u) key[u] = i0 + u < s1[k] ? src[i0 + u] : 0ull;
                const u32 left = s1[k] - i0;
                count(key, one, left < (u32)KU ? left : (u32)KU);
            }
        }
        PHASE_MARK(3);
    }
    __syncthreads();
    // ---- the occupied sub-ranges in order (their ids), the overflow values in ascending order ----
    u64 sk[BSL];
    u32 md = 0;
#pragma unroll
    for (int q = 0; q < BSL; ++q) {
        sk[q] = tk[tid * BSL + q];
        md += sk[q] != EMPTY ? 1u : 0u;
    }
    u32 dm;
    u32 at = bins_incl_scan<u32>(md, wt32, dm) - md;
#pragma unroll
    for (int q = 0; q < BSL; ++q) {
        if (sk[q] != EMPTY) {
            if (at < (u32)MAIN_CAP) ids[at] = (unsigned short)(tid * BSL + q);
            ++at;
        }
    }
    constexpr int OSL = OVS / BW;
    u64 ok2[OSL];
    u32 mo = 0;
#pragma unroll
    for (int q = 0; q < OSL; ++q) {
        ok2[q] = ovk[tid * OSL + q];
        mo += ok2[q] != EMPTY ? 1u : 0u;
    }
    u32 no;
    u32 oat = bins_incl_scan<u32>(mo, wt32, no) - mo;
#pragma unroll
    for (int q = 0; q < OSL; ++q) {
        if (ok2[q] != EMPTY) {
            if (oat < (u32)OVCAP) {
                osk[oat] = ok2[q];
                osc[oat] = ovc[tid * OSL + q];
            }
            ++oat;
        }
    }
    const bool bad = dm > (u32)MAIN_CAP || no > (u32)OVCAP;
    // (one lane reads the error word for all: another workgroup may be setting it at this moment)
    if (__syncthreads_or(bad || (tid == 0 && (over || w.ctl->err)))) {  // the keys do not suit the index, or an earlier failure
        if (tid == 0) {
            if (bad || over) {
                if (!(atomicOr(&w.ctl->err, ERR_BIN) & ERR_BIN)) {  // (the first bin to refuse tells why)
                    w.ctl->bad_bin = b;
                    w.ctl->bad_main = dm;
                    w.ctl->bad_over = no > nov ? no : nov;
                }
            }
            w.bin_d[b] = 0;
            w.bin_c[b] = 0;
            w.bin_w[b] = 0.0;
        }
        return;
    }
    for (u32 i = tid; i < no; i += BW) {  // rank among the overflow values: they are distinct
        const u64 v = osk[i];
        u32 rk = 0;
        for (u32 j2 = 0; j2 < no; ++j2) rk += osk[j2] < v ? 1u : 0u;
        ovk[rk] = v;  // (the table's own memory: every slot was read before the barrier above)
        ovc[rk] = osc[i];
    }
    __syncthreads();
    PHASE_MARK(4);
    // ---- the merged order: a table value's place = table values before it + overflow values below it, and the
    //      other way round (both lists ascend: two searches; no overflow values: the table's order as it is) ----
    for (u32 e = tid; e < dm; e += BW) {
        u32 lo = 0;
        if (no) {
            const u64 v = tk[ids[e]];
            u32 hi = no;  // overflow values below v
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (ovk[mid] < v) lo = mid + 1; else hi = mid;
            }
        }
        ord[e + lo] = ids[e];
    }
    for (u32 i = tid; i < no; i += BW) {
        const u64 v = ovk[i];
        u32 lo = 0, hi = dm;  // table values below v
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (tk[ids[mid]] < v) lo = mid + 1; else hi = mid;
        }
        ord[lo + i] = (unsigned short)(SUB + i);
    }
    __syncthreads();
    PHASE_MARK(5);
    // ---- one pass in ascending order: running copies and weights ----
    const u32 d = dm + no;
    u64 *rv = w.rval + (size_t)b * CAPD;
    u64 *rc = w.rccum + (size_t)b * CAPD;
    double *rw = w.rwcum + (size_t)b * CAPD;
    u64 ccar = 0;
    double wcar = 0.0;
    for (u32 p0 = 0; p0 < d; p0 += BW) {
        const u32 p = p0 + tid;
        u64 v = 0, c = 0;
        double wg = 0.0;
        if (p < d) {
            const u32 id = ord[p];
            v = id < (u32)SUB ? tk[id] : ovk[id - SUB];
            c = id < (u32)SUB ? tc[id] : ovc[id - SUB];
            wg = (double)c * exp2(__longlong_as_double((long long)v) - w.logn);
        }
        u64 ct;
        double wtt;
        const u64 ci = bins_incl_scan<u64>(c, wt64, ct);
        const double wi = bins_incl_scan<double>(wg, wtf, wtt);
        if (p < d) {
            rv[p] = v;
            rc[p] = ccar + ci;
            rw[p] = wcar + wi;
        }
        ccar += ct;
        wcar += wtt;
    }
    if (tid == 0) {
        w.bin_d[b] = d;
        w.bin_c[b] = ccar;
        w.bin_w[b] = wcar;
    }
    PHASE_MARK(6);
    if (w.phase && tid == 0) {  // (development aid: the longest workgroup, and when the last one ended)
        atomicMax(&w.phase[14], t_phase - t_first);
        atomicMax(&w.phase[15], t_phase);
        atomicMin(&w.phase[13], ~t_first);  // (~ of the earliest start: the counters are cleared to zero)
    }
}

// ---- prefix over the bins, runs to their final places ---------------------------------------------------------
__device__ __forceinline__ void rl_final_body(const RWs &w) {
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
        w.out_info[6] = e & ERR_BIN ? ((u64)w.ctl->bad_bin << 32) | w.P : 0ull;
        w.out_info[7] = e & ERR_BIN ? ((u64)w.ctl->bad_main << 32) | w.ctl->bad_over : 0ull;
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


// the four steps as kernels: of one build (mc_rank_runs_build up to 2^26 keys), and of the BIG_G builds a large one
// is made of, side by side (blockIdx.y picks the build; its description lies in memory, written by k_big_setup)
__global__ void __launch_bounds__(SNIFF_THREADS) k_rl_sniff(RWs w) { rl_sniff_body(w); }
__global__ void __launch_bounds__(WG, 4) k_rl_strip(RWs w) { rl_strip_body(w); }  // (4 waves per SIMD = two workgroups a CU)
__global__ void __launch_bounds__(BW) k_rl_bins(RWs w) { rl_bins_body(w); }
__global__ void __launch_bounds__(WG) k_rl_final(RWs w) { rl_final_body(w); }
__global__ void __launch_bounds__(SNIFF_THREADS) k_rlg_sniff(const RWs *pw) {
    const RWs w = pw[blockIdx.y];
    if (w.n) rl_sniff_body(w);
}
__global__ void __launch_bounds__(WG, 4) k_rlg_strip(const RWs *pw) {
    const RWs w = pw[blockIdx.y];
    if (w.n) rl_strip_body(w);
}
__global__ void __launch_bounds__(BW) k_rlg_bins(const RWs *pw) {
    const RWs w = pw[blockIdx.y];
    if (w.n) rl_bins_body(w);
}
__global__ void __launch_bounds__(WG) k_rlg_final(const RWs *pw) {
    const RWs w = pw[blockIdx.y];
    if (w.n) rl_final_body(w);
}


// ======== builds of more than 2^26 keys ==========================================================================
// The same index from two levels of the same steps.  Level 1 (k_rl_sniff over all keys, then k_big_strip) reads every
// key ONCE: copies of the heavy values the sample found are counted in LDS as k_rl_strip counts them; every other key
// goes to one of BIG_G value-range GROUPS (the sample's tail in equal shares: the look-up table at equal depth) --
// buffered in LDS, ordered by group, and written behind the group's cursor, one atomic add per group and batch.  A
// group is then an input like any other, of a few million keys in one value range: k_big_setup writes the
// description of each group's build into memory and the four steps run for all groups side by side (k_rlg_*: the
// bodies above, blockIdx.y = group; a group's own sample finds the values that are frequent INSIDE it -- the second
// tier, ~1e-5 of all keys each -- and its own look-up table spreads ITS range over its bins).  k_big_join puts the
// groups' runs one after the other, the level-1 heavy values in between where they belong, and adds the bases.
// Nothing waits for anything inside a kernel.  A group that outgrows its share of the workspace (keys whose tail is
// most of the input, or falls into few groups) or a bin that outgrows its table fails the build, loudly, as above.
// Where this stands (scripts/runs_big_check.py, profiles/r03_*_runs_two_level.txt): correct against mc_rank_build at
// 7e7 and 2e8 keys of both benchmark models (every count equal, positions within 6e-15), 2.56 against 3.12 ms at 2e8
// keys of the v2 model but 1.75 against 1.49 ms at 7e7; at 1.25e9 keys of the v2 model the last groups hold 0.8 - 1e6
// keys that are ALL distinct -- more than a group's 256 tables take -- and the build refuses after 11 ms (k_big_strip
// 5.6 ms: its tail is 40 % of the keys with the 16384-key sample's list; k_rlg_strip 2.2, k_rlg_bins 2.7).  The host
// classes therefore keep mc_rank_build beyond 2^26 keys (mc_rank_runs_max_keys); callers that pass their own
// RunsBuffers get this build.
constexpr int BIG_G = 64;      // groups
constexpr int BIG_P = 256;     // bins of a group's own build
constexpr u64 BIG_MAX_N = 0xFFFFFFFFull;
enum { ERR_GROUP = 64u };

struct BigCtl {
    u32 err, pad;
    u64 gcur[BIG_G];  // keys written to each group
};

struct BWs {
    RWs top;          // level 1: ctl, heavy, hcnt, in, hbin, lut, cell0; n; P = BIG_G
    RWs templ;        // a group's build, group 0's pointers (group g's: + g * slice bytes), fields for `cap` keys
    size_t slice;
    BigCtl *bctl;
    RWs *problems;    // [BIG_G]
    u64 *gkeys;       // [BIG_G][cap]
    u64 n;            // all keys
    u64 cap;          // keys a group takes
    u32 region;       // keys per workgroup of k_big_strip (whole tiles)
    u64 *g_vals, *g_ccum;  // the groups' own indexes: [BIG_G][gcap]
    double *g_wcum;
    u64 *g_info;      // [BIG_G][8]
    u64 gcap;
    u64 *out_vals, *out_ccum;
    double *out_wcum;
    u64 *out_info;
    u64 out_cap;
};

__global__ void __launch_bounds__(64) k_big_clear(BWs bw) {
    if (threadIdx.x == 0) bw.bctl->err = 0;
    if (threadIdx.x < BIG_G) bw.bctl->gcur[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(WG, 4) k_big_strip(BWs bw) {
    static_assert(BIG_G == 64, "one wave scans the groups' counts");
    const RWs &w = bw.top;
    __shared__ u64 hk[HSET + 1];
    __shared__ u32 hs[HSET + 1];
    __shared__ u32 hi[HSET];
    __shared__ u64 tb[TBUF];
    __shared__ unsigned short slut[CELLS];
    __shared__ u32 gh[BIG_G], goff[BIG_G + 1];
    __shared__ u64 gbase[BIG_G];
    __shared__ u32 tfill;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 H = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
    const BinMap bm = bin_map(w.ctl->map_lo, w.ctl->map_hi, slut);
    for (u32 i = tid; i <= (u32)HSET; i += WG) {
        hk[i] = EMPTY;
        hs[i] = 0;
    }
    for (u32 i = tid; i < (u32)CELLS / 2; i += WG) ((u32 *)slut)[i] = ((const u32 *)w.lut)[i];
    __syncthreads();
    for (u32 i = tid; i < H; i += WG) {
        const u64 v = w.heavy[i];
        if (v == EMPTY) continue;
        u32 s = strip_hash(v);
        for (;;) {
            const u64 old = atomicCAS(&hk[s], EMPTY, v);
            if (old == EMPTY) {
                hi[s] = i;
                break;
            }
            if (old == v) break;
            s = (s + 1) & (HSET - 1);
        }
    }
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;

    // one batch: the first m keys of tb, ordered by group, each group's share written behind its cursor
    auto flush = [&](u32 m) {
        if (tid < BIG_G) gh[tid] = 0;
        __syncthreads();
        u64 key[ITEMS];
        u32 grp[ITEMS], rk[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            key[q] = i < m ? tb[i] : 0ull;
            grp[q] = bin_of(bm, key[q]);
            if (grp[q] >= (u32)BIG_G) grp[q] = BIG_G - 1;
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            rk[q] = i < m ? atomicAdd(&gh[grp[q]], 1u) : 0u;
        }
        __syncthreads();
        if (tid < BIG_G) {
            const u32 c = gh[tid];
            u32 inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            goff[tid] = inc - c;
            if (tid == BIG_G - 1) goff[BIG_G] = inc;
            gbase[tid] = c ? atomicAdd(&bw.bctl->gcur[tid], (u64)c) : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 i = q * WG + tid;
            if (i < m) {
                const u32 at = goff[grp[q]] + rk[q];
                if (at < m) tb[at] = key[q]; else atomicOr(&bw.bctl->err, ERR_RANGE);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {  // consecutive lanes on consecutive addresses inside a group's share
            const u32 i = q * WG + tid;
            if (i < m) {
                const u64 k2 = tb[i];
                u32 g = bin_of(bm, k2);
                if (g >= (u32)BIG_G) g = BIG_G - 1;
                const u64 at = gbase[g] + (i - goff[g]);
                if (at < bw.cap) bw.gkeys[(u64)g * bw.cap + at] = k2; else atomicOr(&bw.bctl->err, ERR_GROUP);
            }
        }
        __syncthreads();
    };

    const u64 n = bw.n;
    const u64 lo = (u64)blockIdx.x * bw.region;
    if (lo < n) {
        const u32 rn = (u32)(n - lo < (u64)bw.region ? n - lo : (u64)bw.region);
        if (tid == 0) tfill = 0;
        __syncthreads();
        u64 k[ITEMS], nk[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 j = q * WG + tid;
            nk[q] = w.in[lo + (j < rn ? j : rn - 1)];
        }
        for (u32 c0 = 0; c0 < rn; c0 += TILE) {
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) k[q] = nk[q];
            if (c0 + TILE < rn) {
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = c0 + TILE + q * WG + tid;
                    nk[q] = w.in[lo + (j < rn ? j : rn - 1)];
                }
            }
            u64 tmask[ITEMS];
            u32 total = 0;
            {
                u32 sl[ITEMS];
                u64 a[ITEMS], b[ITEMS];
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    sl[q] = strip_hash(k[q]);
                    a[q] = hk[sl[q]];
                    b[q] = hk[sl[q] + 1];
                }
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const bool valid = c0 + q * WG + tid < rn;
                    const bool real = valid && k[q] != EMPTY;  // (a key of all ones matches the table's empty marker: never heavy)
                    const bool hit_a = real && a[q] == k[q];
                    const bool hit_b = real && !hit_a && a[q] != EMPTY && b[q] == k[q];
                    const bool walk = real && !hit_a && !hit_b && a[q] != EMPTY && (b[q] != EMPTY || sl[q] + 1 == (u32)HSET);
                    bool heavy = hit_a || hit_b;
                    u32 slot = sl[q] + (hit_b ? 1u : 0u);
                    if (__ballot(walk)) {  // (rare)
                        if (walk) {
                            u32 t = sl[q] + 1 == (u32)HSET ? 0u : (sl[q] + 2 == (u32)HSET ? 0u : sl[q] + 2);
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
                    tmask[q] = __ballot(valid && !heavy);
                    total += (u32)__popcll(tmask[q]);
                }
            }
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
                flush((u32)TBUF);
#pragma unroll
                for (int q = 0; q < ITEMS; ++q)
                    if (pos[q] != 0xFFFFFFFFu && pos[q] >= (u32)TBUF) tb[pos[q] - (u32)TBUF] = k[q];
                if (tid == 0) tfill -= (u32)TBUF;
                __syncthreads();
            }
        }
        const u32 m = tfill;
        __syncthreads();
        if (m) flush(m);
    }
    __syncthreads();
    u32 *hc = (u32 *)tb;
    for (u32 i = tid; i < H; i += WG) hc[i] = 0;
    __syncthreads();
    for (u32 i = tid; i < (u32)HSET; i += WG)
        if (hk[i] != EMPTY && hs[i]) hc[hi[i]] = hs[i];
    __syncthreads();
    for (u32 i = tid; i < H; i += WG)
        if (hc[i]) atomicAdd(&w.hcnt[i], hc[i]);
}

template <typename T>
__device__ __forceinline__ void move_on(T *&p, size_t by) { p = (T *)((char *)p + by); }

// the description of every group's own build: group 0's with every pointer moved on by g slices, the keys the
// group received, its regions
__global__ void __launch_bounds__(64) k_big_setup(BWs bw) {
    const u32 g = threadIdx.x;
    if (g >= (u32)BIG_G) return;
    const u64 ng = bw.bctl->gcur[g];
    const bool fits = ng <= bw.cap;
    if (!fits) atomicOr(&bw.bctl->err, ERR_GROUP);
    RWs p = bw.templ;
    const size_t by = (size_t)g * bw.slice;
    move_on(p.ctl, by); move_on(p.heavy, by); move_on(p.hcnt, by); move_on(p.tail, by); move_on(p.nbatch, by);
    move_on(p.htab, by); move_on(p.hidx, by);
    move_on(p.rmax, by); move_on(p.hbin, by); move_on(p.lut, by); move_on(p.cell0, by); move_on(p.start, by);
    move_on(p.rval, by); move_on(p.rccum, by); move_on(p.rwcum, by); move_on(p.bin_d, by); move_on(p.bin_c, by);
    move_on(p.bin_w, by);
    p.in = bw.gkeys + (u64)g * bw.cap;
    p.n = fits ? (u32)ng : 0u;
    p.regions = p.n ? (p.n + p.region - 1) / p.region : 0u;
    p.out_vals = bw.g_vals + (u64)g * bw.gcap;
    p.out_ccum = bw.g_ccum + (u64)g * bw.gcap;
    p.out_wcum = bw.g_wcum + (u64)g * bw.gcap;
    p.out_info = bw.g_info + (u64)g * 8;
    p.out_cap = bw.gcap;
    bw.problems[g] = p;
    if (!p.n)
        for (int i = 0; i < 8; ++i) p.out_info[i] = 0;  // (an index of no runs; k_rlg_* leave the group alone)
}

// the groups' runs in order, the level-1 heavy values where they belong, the running counts and weights on top of
// what lies before.  Every workgroup adds up the same bases in the same order.
__global__ void __launch_bounds__(256) k_big_join(BWs bw) {
    const RWs &w = bw.top;
    __shared__ u64 hv[H_MAX];
    __shared__ u64 hcp[H_MAX + 1];
    __shared__ double hwp[H_MAX + 1];
    __shared__ u64 dbase[BIG_G + 1], cbase[BIG_G + 1];
    __shared__ double wbase[BIG_G + 1];
    __shared__ u32 s_err;
    __shared__ u64 s_heavy2, s_tail2;
    const int tid = threadIdx.x;
    const u32 H = w.ctl->n_heavy <= (u32)H_MAX ? w.ctl->n_heavy : 0u;
    for (u32 i = tid; i < H; i += 256) hv[i] = w.heavy[i];
    if (tid == 0) {
        u32 e = bw.bctl->err | w.ctl->err;
        u64 d = 0, c = 0, h2 = 0, t2 = 0;
        double ws = 0.0;
        for (int g = 0; g < BIG_G; ++g) {
            const u64 *gi = bw.g_info + (u64)g * 8;
            dbase[g] = d;
            cbase[g] = c;
            wbase[g] = ws;
            const u64 dg = gi[0] <= bw.gcap ? gi[0] : 0ull;
            e |= (u32)gi[1];
            if (gi[2] != bw.bctl->gcur[g]) e |= ERR_COUNT;  // (every key a group received is in its index)
            h2 += gi[3];
            t2 += gi[4];
            if (dg) {
                c += bw.g_ccum[(u64)g * bw.gcap + dg - 1];
                ws += bw.g_wcum[(u64)g * bw.gcap + dg - 1];
            }
            d += dg;
        }
        dbase[BIG_G] = d;
        cbase[BIG_G] = c;
        wbase[BIG_G] = ws;
        u64 hc = 0;
        double hw = 0.0;
        hcp[0] = 0;
        hwp[0] = 0.0;
        for (u32 i = 0; i < H; ++i) {
            const u64 cnt = w.hcnt[i];
            hc += cnt;
            hw += (double)cnt * exp2(__longlong_as_double((long long)w.heavy[i]) - w.logn);
            hcp[i + 1] = hc;
            hwp[i + 1] = hw;
        }
        if (!e && c + hc != bw.n) e |= ERR_COUNT;
        if (!e && d + H > bw.out_cap) e |= ERR_RANGE;
        s_err = e;
        s_heavy2 = h2;
        s_tail2 = t2;
    }
    __syncthreads();
    const u32 err = s_err;
    const u64 D2 = dbase[BIG_G];
    if (blockIdx.x == 0 && tid == 0) {
        bw.out_info[0] = err ? 0ull : D2 + H;
        bw.out_info[1] = err;
        bw.out_info[2] = cbase[BIG_G] + hcp[H];
        bw.out_info[3] = H + s_heavy2;
        bw.out_info[4] = s_tail2;
        bw.out_info[5] = 0;
        // (what refused: a group's keys against its share, or the first refusing group's own words)
        bw.out_info[6] = 0;
        bw.out_info[7] = 0;
        if (err & ERR_GROUP) {
            u64 most = 0;
            for (int g = 0; g < BIG_G; ++g) most = bw.bctl->gcur[g] > most ? bw.bctl->gcur[g] : most;
            bw.out_info[6] = most;
            bw.out_info[7] = bw.cap;
        } else if (err & ERR_BIN) {
            for (int g = 0; g < BIG_G; ++g)
                if (bw.g_info[(u64)g * 8 + 1] & ERR_BIN) {
                    bw.out_info[6] = bw.g_info[(u64)g * 8 + 6] | ((u64)g << 56);
                    bw.out_info[7] = bw.g_info[(u64)g * 8 + 7];
                    break;
                }
        }
    }
    if (err) return;
    const u64 stride = (u64)gridDim.x * 256u;
    for (u64 j = (u64)blockIdx.x * 256u + tid; j < D2; j += stride) {  // a run of a group
        u32 g = 0;
        for (u32 st = BIG_G / 2; st; st >>= 1)
            if (dbase[g + st] <= j) g += st;
        const u64 l = j - dbase[g], at = (u64)g * bw.gcap + l;
        const u64 v = bw.g_vals[at];
        u32 lo = 0, hi = H;  // level-1 heavy values below v
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (hv[mid] < v) lo = mid + 1; else hi = mid;
        }
        const u64 o = j + lo;
        bw.out_vals[o] = v;
        bw.out_ccum[o] = cbase[g] + bw.g_ccum[at] + hcp[lo];
        bw.out_wcum[o] = (wbase[g] + bw.g_wcum[at]) + hwp[lo];
    }
    if (blockIdx.x == 0) {
        const BinMap bm = bin_map(w.ctl->map_lo, w.ctl->map_hi, w.lut);
        for (u32 i = tid; i < H; i += 256) {  // a level-1 heavy value: behind the runs of its group that lie below it
            const u64 v = hv[i];
            u32 g = bin_of(bm, v);
            if (g >= (u32)BIG_G) g = BIG_G - 1;
            const u64 dg = dbase[g + 1] - dbase[g];
            const u64 *gv = bw.g_vals + (u64)g * bw.gcap;
            u64 lo = 0, hi = dg;
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (gv[mid] < v) lo = mid + 1; else hi = mid;
            }
            const u64 o = dbase[g] + lo + i;
            const u64 cb = cbase[g] + (lo ? bw.g_ccum[(u64)g * bw.gcap + lo - 1] : 0ull);
            const double wb = lo ? wbase[g] + bw.g_wcum[(u64)g * bw.gcap + lo - 1] : wbase[g];
            bw.out_vals[o] = v;
            bw.out_ccum[o] = cb + hcp[i + 1];
            bw.out_wcum[o] = wb + hwp[i + 1];
        }
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

// development aid: per-phase clock sums (mc_rank_runs_debug_phases)
__device__ u64 g_phase[16];
std::atomic<int> g_phase_on{0};
u64 *phase_ptr() {
    void *p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase)) == hipSuccess ? (u64 *)p : nullptr;
}

struct RLayout {
    size_t ctl, heavy, hcnt, htab, hidx, nbatch, rmax, hbin, lut, cell0, bin_d, bin_c, bin_w, start, rval, rccum, rwcum, tail, end;
    u32 region, regions, nb, P;
};

RLayout layout_for(u64 n, u32 force_P = 0) {
    RLayout L;
    // keys per strip region: whole tiles, as few as give every k_rl_strip workgroup the GPU holds at once one region
    const u64 tiles = (n + TILE - 1) / TILE;
    const u64 per = (tiles + STRIP_SLOTS - 1) / STRIP_SLOTS;
    L.region = (u32)((per ? per : 1) * TILE);
    L.regions = (u32)((n + L.region - 1) / L.region);
    if (L.regions == 0) L.regions = 1;
    L.nb = L.region / TBUF;
    // bins: a bin costs a workgroup of k_rl_bins whatever it holds, and a bin may hold CAPD distinct values
    L.P = force_P ? force_P : n >= (1ull << 21) ? 1024u : n >= (1ull << 17) ? 256u : 64u;
    size_t p = 0;
    L.ctl = p; p += align256(sizeof(RCtl));
    L.heavy = p; p += align256(H_MAX * sizeof(u64));
    L.hcnt = p; p += align256(H_MAX * sizeof(u32));
    L.htab = p; p += align256(HSET * sizeof(u64));
    L.hidx = p; p += align256(HSET * sizeof(u32));
    L.nbatch = p; p += align256((size_t)L.regions * sizeof(u32));
    L.rmax = p; p += align256((size_t)L.regions * sizeof(u64));
    L.hbin = p; p += align256((P_MAX + 1) * sizeof(u32));
    L.lut = p; p += align256(CELLS * sizeof(unsigned short));
    L.cell0 = p; p += align256((P_MAX + 1) * sizeof(unsigned short));
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

struct BLayout {
    size_t ctl, heavy, hcnt, hbin, lut, cell0, bctl, problems, g_info, g_vals, g_ccum, g_wcum, gkeys, slices, slice, end;
    u64 cap, gcap;
    u32 region, regions;
    RLayout grp;
};

BLayout big_layout(u64 n) {
    BLayout B;
    const u64 tiles = (n + TILE - 1) / TILE;
    const u64 per = (tiles + STRIP_SLOTS - 1) / STRIP_SLOTS;
    B.region = (u32)(per * TILE);
    B.regions = (u32)((n + B.region - 1) / B.region);
    // a group takes 1.25 n / BIG_G keys (whole tiles): the tail of keys that suit the index is under half of them and
    // the groups share it half by keys, half by distinct values; whatever needs more is refused (ERR_GROUP) and
    // sorted by mc_rank_build
    B.cap = (((n + n / 4) / BIG_G + TILE - 1) / TILE + 1) * TILE;
    B.gcap = (u64)BIG_P * CAPD + 1;
    B.grp = layout_for(B.cap, BIG_P);
    B.slice = align256(B.grp.end);
    size_t p = 0;
    B.ctl = p; p += align256(sizeof(RCtl));
    B.heavy = p; p += align256(H_MAX * sizeof(u64));
    B.hcnt = p; p += align256(H_MAX * sizeof(u32));
    B.hbin = p; p += align256((P_MAX + 1) * sizeof(u32));
    B.lut = p; p += align256(CELLS * sizeof(unsigned short));
    B.cell0 = p; p += align256((P_MAX + 1) * sizeof(unsigned short));
    B.bctl = p; p += align256(sizeof(BigCtl));
    B.problems = p; p += align256(BIG_G * sizeof(RWs));
    B.g_info = p; p += align256(BIG_G * 8 * sizeof(u64));
    B.g_vals = p; p += align256((size_t)BIG_G * B.gcap * sizeof(u64));
    B.g_ccum = p; p += align256((size_t)BIG_G * B.gcap * sizeof(u64));
    B.g_wcum = p; p += align256((size_t)BIG_G * B.gcap * sizeof(double));
    B.gkeys = p; p += align256((size_t)BIG_G * B.cap * sizeof(u64));
    B.slices = p; p += (size_t)BIG_G * B.slice;
    B.end = p;
    return B;
}

int build_big(const double *keys_in, uint64_t n, uint64_t n_global, double *vals, uint64_t *cum_count, double *cum_pos,
              uint64_t capacity, uint64_t *info, void *workspace, size_t workspace_bytes, hipStream_t s) {
    const BLayout B = big_layout(n);
    if (workspace_bytes < B.end) {
        mc_set_error("mc_rank_runs_build: workspace of %zu bytes, need %zu", workspace_bytes, B.end);
        return MC_ERR_WORKSPACE;
    }
    if (capacity < mc_rank_runs_capacity(n)) {
        mc_set_error("mc_rank_runs_build: output arrays of %llu entries, need %llu (mc_rank_runs_capacity)",
                     (unsigned long long)capacity, (unsigned long long)mc_rank_runs_capacity(n));
        return MC_ERR_ARG;
    }
    char *base = (char *)workspace;
    BWs bw;
    std::memset(&bw, 0, sizeof bw);
    const double logn = std::log2((double)n_global);
    RWs &t = bw.top;
    t.ctl = (RCtl *)(base + B.ctl);
    t.heavy = (u64 *)(base + B.heavy);
    t.hcnt = (u32 *)(base + B.hcnt);
    t.hbin = (u32 *)(base + B.hbin);
    t.lut = (unsigned short *)(base + B.lut);
    t.cell0 = (unsigned short *)(base + B.cell0);
    t.in = (const u64 *)keys_in;
    t.n = (u32)n;
    t.P = BIG_G;
    t.logn = logn;
    t.eq_share = 0.5;      // groups: half by keys (a group's share of the workspace), half by distinct values (its bins' tables)
    t.eq_distinct = 0.5;
    char *g0 = base + B.slices;
    const RLayout &L = B.grp;
    RWs &w = bw.templ;
    w.ctl = (RCtl *)(g0 + L.ctl);
    w.heavy = (u64 *)(g0 + L.heavy);
    w.hcnt = (u32 *)(g0 + L.hcnt);
    w.htab = (u64 *)(g0 + L.htab);
    w.hidx = (u32 *)(g0 + L.hidx);
    w.tail = (u64 *)(g0 + L.tail);
    w.nbatch = (u32 *)(g0 + L.nbatch);
    w.rmax = (u64 *)(g0 + L.rmax);
    w.hbin = (u32 *)(g0 + L.hbin);
    w.lut = (unsigned short *)(g0 + L.lut);
    w.cell0 = (unsigned short *)(g0 + L.cell0);
    w.start = (u32 *)(g0 + L.start);
    w.rval = (u64 *)(g0 + L.rval);
    w.rccum = (u64 *)(g0 + L.rccum);
    w.rwcum = (double *)(g0 + L.rwcum);
    w.bin_d = (u32 *)(g0 + L.bin_d);
    w.bin_c = (u64 *)(g0 + L.bin_c);
    w.bin_w = (double *)(g0 + L.bin_w);
    w.region = L.region;
    w.nb = L.nb;
    w.P = L.P;
    w.logn = logn;
    w.eq_share = 0.0;      // a group's bins: by distinct values, a quarter by place
    w.eq_distinct = 0.75;
    bw.slice = B.slice;
    bw.bctl = (BigCtl *)(base + B.bctl);
    bw.problems = (RWs *)(base + B.problems);
    bw.gkeys = (u64 *)(base + B.gkeys);
    bw.cap = B.cap;
    bw.region = B.region;
    bw.g_vals = (u64 *)(base + B.g_vals);
    bw.g_ccum = (u64 *)(base + B.g_ccum);
    bw.g_wcum = (double *)(base + B.g_wcum);
    bw.g_info = (u64 *)(base + B.g_info);
    bw.gcap = B.gcap;
    bw.n = n;
    bw.out_vals = (u64 *)vals;
    bw.out_ccum = (u64 *)cum_count;
    bw.out_wcum = cum_pos;
    bw.out_info = (u64 *)info;
    bw.out_cap = capacity;
    const u32 gx = L.regions < (u32)STRIP_SLOTS ? L.regions : (u32)STRIP_SLOTS;
    hipLaunchKernelGGL(k_big_clear, dim3(1), dim3(64), 0, s, bw);
    hipLaunchKernelGGL(k_rl_sniff, dim3(1), dim3(SNIFF_THREADS), 0, s, bw.top);
    hipLaunchKernelGGL(k_big_strip, dim3(B.regions), dim3(WG), 0, s, bw);
    hipLaunchKernelGGL(k_big_setup, dim3(1), dim3(64), 0, s, bw);
    hipLaunchKernelGGL(k_rlg_sniff, dim3(1, BIG_G), dim3(SNIFF_THREADS), 0, s, (const RWs *)bw.problems);
    hipLaunchKernelGGL(k_rlg_strip, dim3(gx, BIG_G), dim3(WG), 0, s, (const RWs *)bw.problems);
    hipLaunchKernelGGL(k_rlg_bins, dim3(L.P, BIG_G), dim3(BW), 0, s, (const RWs *)bw.problems);
    hipLaunchKernelGGL(k_rlg_final, dim3(L.P, BIG_G), dim3(WG), 0, s, (const RWs *)bw.problems);
    hipLaunchKernelGGL(k_big_join, dim3(2048), dim3(256), 0, s, bw);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

}  // namespace

extern "C" size_t mc_rank_runs_workspace_bytes(uint64_t n) {
    if (n == 0 || n > BIG_MAX_N) return 0;
    return n <= RUNS_MAX_N ? layout_for(n).end : big_layout(n).end;
}

// (the size up to which the index is the default of the host classes; mc_rank_runs_build itself takes up to 2^32 - 1
// keys, in two levels beyond this -- measured slower than mc_rank_build at 7e7 keys and refused at 1.25e9, see above)
extern "C" uint64_t mc_rank_runs_max_keys(void) { return RUNS_MAX_N; }

extern "C" uint64_t mc_rank_runs_capacity(uint64_t n) {
    if (n == 0 || n > BIG_MAX_N) return 0;
    if (n > RUNS_MAX_N) return (u64)BIG_G * big_layout(n).gcap + H_MAX + 1;
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
    if (n > BIG_MAX_N) {
        mc_set_error("mc_rank_runs_build: %llu keys, the run index takes at most %llu (use mc_rank_build beyond that)",
                     (unsigned long long)n, (unsigned long long)BIG_MAX_N);
        return MC_ERR_ARG;
    }
    if (n > RUNS_MAX_N)
        return build_big(keys_in, n, n_global, vals, cum_count, cum_pos, capacity, info, workspace, workspace_bytes, s);
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
    w.htab = (u64 *)(base + L.htab);
    w.hidx = (u32 *)(base + L.hidx);
    w.in = (const u64 *)keys_in;
    w.tail = (u64 *)(base + L.tail);
    w.nbatch = (u32 *)(base + L.nbatch);
    w.rmax = (u64 *)(base + L.rmax);
    w.hbin = (u32 *)(base + L.hbin);
    w.lut = (unsigned short *)(base + L.lut);
    w.cell0 = (unsigned short *)(base + L.cell0);
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
    w.eq_share = 0.25;
    w.eq_distinct = 0.0;
    static const int env_reuse = getenv("MC_SORT_REUSE") ? atoi(getenv("MC_SORT_REUSE")) : 1;
    w.reuse_list = env_reuse ? 1u : 0u;
    w.debug_fail = 0;
    w.out_vals = (u64 *)vals;
    w.out_ccum = (u64 *)cum_count;
    w.out_wcum = cum_pos;
    w.out_info = (u64 *)info;
    w.out_cap = capacity;
    w.phase = g_phase_on.load() ? phase_ptr() : nullptr;
    hipLaunchKernelGGL(k_rl_sniff, dim3(1), dim3(SNIFF_THREADS), 0, s, w);
    hipLaunchKernelGGL(k_rl_strip, dim3(L.regions < (u32)STRIP_SLOTS ? L.regions : (u32)STRIP_SLOTS), dim3(WG), 0, s, w);
    hipLaunchKernelGGL(k_rl_bins, dim3(L.P), dim3(BW), 0, s, w);
    hipLaunchKernelGGL(k_rl_final, dim3(L.P), dim3(WG), 0, s, w);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_rank_runs_forget(void *workspace, uint64_t n, void *stream) {
    if (!workspace || n == 0 || n > BIG_MAX_N) {
        mc_set_error("mc_rank_runs_forget: null workspace or a key count outside 1 .. 2^32 - 1");
        return MC_ERR_ARG;
    }
    HIP_TRY(hipMemsetAsync((char *)workspace + (n <= RUNS_MAX_N ? layout_for(n).ctl : big_layout(n).ctl), 0, sizeof(RCtl),
                           (hipStream_t)stream));
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
    if (local[1] & (ERR_BIN | ERR_GROUP)) {
        mc_set_error("mc_rank_runs_build: these keys do not suit the run index (a bin of the value range holds more "
                     "than %d distinct values, or a value-range group of a large build more than its share of the "
                     "workspace); queries answer NaN -- build with mc_rank_build instead", CAPD);
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

// Development aid (scripts/runs_big_check.py): what the groups of the last large build in `workspace` received and
// made of it: out[g * 9] = keys, out[g * 9 + 1 .. 8] = the group's own info words.
extern "C" int mc_rank_runs_debug_groups(const void *workspace, uint64_t n, uint64_t *out, void *stream) {
    if (!workspace || !out || n <= RUNS_MAX_N || n > BIG_MAX_N) {
        mc_set_error("mc_rank_runs_debug_groups: bad argument");
        return MC_ERR_ARG;
    }
    const BLayout B = big_layout(n);
    BigCtl c;
    uint64_t gi[BIG_G * 8];
    HIP_TRY(hipMemcpyAsync(&c, (const char *)workspace + B.bctl, sizeof c, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(gi, (const char *)workspace + B.g_info, sizeof gi, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int g = 0; g < BIG_G; ++g) {
        out[g * 9] = c.gcur[g];
        for (int i = 0; i < 8; ++i) out[g * 9 + 1 + i] = gi[g * 8 + i];
    }
    return MC_OK;
}

// Development aid (scripts/runs_prof.py): on != 0 makes the builds that follow add their workgroups' per-phase
// clock ticks (100 MHz) to 16 device counters; out (may be NULL) receives the counters, which are cleared.
extern "C" int mc_rank_runs_debug_phases(int on, uint64_t out[16]) {
    u64 *p = phase_ptr();
    if (!p) {
        mc_set_error("mc_rank_runs_debug_phases: no device symbol");
        return MC_ERR_HIP;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (out) HIP_TRY(hipMemcpy(out, p, 16 * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(p, 0, 16 * sizeof(u64)));
    g_phase_on.store(on ? 1 : 0);
    return MC_OK;
}
