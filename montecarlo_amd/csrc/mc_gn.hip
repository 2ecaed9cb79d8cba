// Guess numbers for a scored test set (gfx950): the parts of MonteCarlo.parse_file and
// MonteCarloLib.ml2p_iter2gc that are per-password loops in the reference,
//     lib4mc/MonteCarloParent.py:68    sorted(..., key=lambda x: x[2])              -> mc_order_keys
//     lib4mc/MonteCarloLib.py:47-55    rank = ceil(max(estimate, prev_rank + add))  -> mc_guess_numbers
//
//   order     stable argsort of the test passwords' -log2 p with the keys-only sort of csrc/mc_sort.hip, twice:
//             the keys (as words whose unsigned order is the doubles' order, both zeros alike) are sorted, every
//             password looks up the index of the first key equal to its own, and the words (that index << 32 |
//             password index) are sorted: their low halves are the order, ties by password index -- what
//             Python's stable sort gives.
//   ranks     in that order rank_i = max(ceil(est_i), rank_{i-1} + add) is i * add + a running maximum of
//             ceil(est_i) - i * add: an inclusive max-scan in int64, exact while ceil(est) < 2^62.  Estimates
//             ascend, so whatever lies beyond that (guess numbers reach 2^90 and more) is a suffix: the kernel
//             reports where it starts and the host finishes it with Python integers, as the reference computes all.
#include <hip/hip_runtime.h>

#include <cmath>

#include "mc_internal.h"
#include "mc_sort.h"

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
typedef long long i64;

// (tiles of 1024 and two searching workgroups a chunk: 1e6 passwords are 977 scanning and 1954 searching workgroups -- the
// step 0.607 ms against 0.611 with tiles of 2048 and four, 0.608-0.610 with 512 / 1024 / 1536 and one or two)
constexpr int GN_THREADS = 256, GN_ITEMS = 4, GN_TILE = GN_THREADS * GN_ITEMS, GN_MAX_BLOCKS = 1024;
constexpr i64 GN_NEG = -(1ll << 62);
constexpr int GN_SUB = 2;  // workgroups a chunk in k_query_chunk_max (a search wants more lanes in flight than a scan)

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// fp64 bits -> a word whose unsigned order is the numeric order (-0.0 and 0.0 compare equal in Python: one word)
__device__ __forceinline__ u64 order_bits(u64 b) { return mc_sort_order_word(b); }

// words[p] = (index of the first sorted key equal to x[p]'s) << 32 | p.  The search runs in two stages: every
// OW_STRIDE-th sorted key is staged in LDS and searched there, the rest of the search is confined to one stride of
// the sorted array (20 dependent reads from memory per password were most of this kernel's time).
// (1024 staged keys and 1024 lanes a workgroup: two passwords a lane, one after the other, instead of eight, and a
// quarter of the scattered reads that stage the keys -- the step 0.599 ms against 0.610 with 4096 / 256; 512 / 512
// 0.5995, 1024 / 512 0.602, 256 / 1024 0.602, 4096 / 512 0.605)
constexpr int OW_THREADS = 1024, OW_LDS = 1024;
constexpr int LS_THREADS = 512, LS_ROUNDS = 4, LS_TILE = LS_THREADS * LS_ROUNDS, LS_CHUNKS = LS_TILE / 64;
constexpr u64 LS_MAX_T = 1ull << 24;
static_assert(LS_TILE % OW_THREADS == 0, "k_order_words walks the LSD passes' tiles");

// the fp64 bits an order word stands for (both zeros come back as +0.0)
__device__ __forceinline__ u64 order_bits_back(u64 w) { return (w >> 63) ? (w & ~(1ull << 63)) : ~w; }

// WORDS: 64-bit words for a second key sort; else only the 32-bit index L, for the LSD passes.  values (or null):
// the sorted scores as numbers, values[i] = the i-th smallest (NaN everywhere if the key sort gave up) -- what a
// caller can put its queries with while the order of the passwords is still being finished (mc_order_keys_begin)
template <bool WORDS>
__global__ void __launch_bounds__(OW_THREADS) k_order_words(const u64 *__restrict__ sorted, const u64 *__restrict__ x,
                                                            u64 t, u32 stride, u64 *__restrict__ words,
                                                            const u32 *sort_err, u32 *failed, u64 *__restrict__ values,
                                                            u32 *__restrict__ table, u32 tiles, u32 mask0) {
    __shared__ u64 top[OW_LDS];
    __shared__ u32 h[256];  // (table != null) the tile's digit histogram of the first LSD pass: k_lsd_hist's, for nothing
    const bool bad = *sort_err != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *failed = bad ? 1u : 0u;  // (the second sort clears the sort's own word)
    const u32 nt = (u32)((t + stride - 1) / stride);  // <= OW_LDS by the caller's choice of stride
    for (u32 i = threadIdx.x; i < nt; i += OW_THREADS) top[i] = sorted[(u64)i * stride];
    __syncthreads();
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {  // (the LSD passes' tiles: LS_TILE passwords in index order)
        if (table) {
            if (threadIdx.x < 256) h[threadIdx.x] = 0;
            __syncthreads();
        }
#pragma unroll 2
        for (int r = 0; r < LS_TILE / OW_THREADS; ++r) {
            const u64 p = (u64)tile * LS_TILE + (u64)r * OW_THREADS + threadIdx.x;
            if (p >= t) break;
            if (values) values[p] = bad ? 0x7FF8000000000000ull : order_bits_back(sorted[p]);
            const u64 v = order_bits(x[p]);
            // the last staged key below v: the first key >= v lies after it, at most one stride further
            u32 a = 0, b = nt;  // first staged index with top[i] >= v
            while (a < b) {
                const u32 mid = (a + b) >> 1;
                if (top[mid] < v) a = mid + 1; else b = mid;
            }
            u64 lo = a ? (u64)(a - 1) * stride + 1 : 0, hi = (u64)a * stride < t ? (u64)a * stride : t;
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (sorted[mid] < v) lo = mid + 1; else hi = mid;
            }
            if (WORDS) words[p] = (lo << 32) | p;
            else ((u32 *)words)[p] = (u32)lo;
            if (table) atomicAdd(&h[(u32)lo & mask0], 1u);
        }
        if (table) {
            __syncthreads();
            if (threadIdx.x < 256) table[(u64)threadIdx.x * tiles + tile] = h[threadIdx.x];  // (its own entry: zeroed by the same thread)
        }
    }
}

// ---- stable LSD radix passes over (L, index) ---------------------------------------------------------------
// The second half of the stable order for up to 2^24 passwords: the pairs (L = index of the first sorted key equal
// to the password's, password index) are sorted by L with the indices kept in their order -- three passes of up to
// 8 bits, each three launches and no workgroup waiting for another: per-tile digit histograms, their prefix over the
// tiles, and a scatter that ranks a tile's elements in index order (a wave finds the lanes that share its digit
// with one ballot per digit bit; the waves' 64-element chunks of a tile are prefixed in LDS).  A keys-only sort of
// the words `L << 32 | index` does the same (and still does above 2^24) at three times the cost: it cannot use
// that the indices arrive in order.

__global__ void __launch_bounds__(LS_THREADS) k_lsd_hist(const u32 *__restrict__ key, u64 t, u32 shift, u32 mask,
                                                         u32 tiles, u32 *__restrict__ table) {
    __shared__ u32 h[256];
    const u32 tile = blockIdx.x;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LS_ROUNDS; ++r) {
        const u64 j = (u64)tile * LS_TILE + r * LS_THREADS + threadIdx.x;
        if (j < t) atomicAdd(&h[(key[j] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) table[(u64)threadIdx.x * tiles + tile] = h[threadIdx.x];
}

// table[bin][tile] -> exclusive prefix over the tiles of the bin, totals[bin] = the bin's size (one workgroup a bin)
__global__ void __launch_bounds__(256) k_lsd_scan(u32 *__restrict__ table, u32 tiles, u32 *__restrict__ totals) {
    __shared__ u32 wsum[4];
    u32 *row = table + (u64)blockIdx.x * tiles;
    const u32 per = (tiles + 255) / 256;
    const u32 lo = threadIdx.x * per < tiles ? threadIdx.x * per : tiles, hi = lo + per < tiles ? lo + per : tiles;
    u32 sum = 0;
    for (u32 i = lo; i < hi; ++i) sum += row[i];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    u32 off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wid) off += wsum[w];
        tot += wsum[w];
    }
    u32 run = off + inc - sum;
    for (u32 i = lo; i < hi; ++i) {
        const u32 v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(LS_THREADS) k_lsd_scatter(const u32 *__restrict__ key, const u32 *__restrict__ pay,
                                                            u64 t, u32 shift, u32 mask, int bits, u32 tiles,
                                                            const u32 *__restrict__ table, const u32 *__restrict__ totals,
                                                            u32 *__restrict__ key_out, u32 *__restrict__ pay_out,
                                                            const u64 *__restrict__ x, u64 *__restrict__ x_sorted,
                                                            const u32 *failed) {
    // (the last pass: x is given, pay_out is the caller's `order` and x_sorted receives the caller's own bits of every
    // key in that order -- a pass of their own before; a key sort that gave up marks both, as k_order_out32 did)
    const bool last = x != nullptr;
    const bool bad = last && *failed != 0;
    __shared__ unsigned short cc[LS_CHUNKS][256];  // elements of each 64-element chunk per digit, then their prefix
    __shared__ u32 gb[256];
    __shared__ u32 wsum[LS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 tile = blockIdx.x;
    for (int i = tid; i < LS_CHUNKS * 256 / 2; i += LS_THREADS) ((u32 *)cc)[i] = 0;
    {   // first place of every digit in the output + this tile's places after the earlier tiles'
        const u32 v = tid <= (int)mask ? totals[tid] : 0u;  // (mask <= 255: the digits in use; k_lsd_scan wrote no other total)
        u32 inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        u32 off = 0;
        for (int w = 0; w < wid; ++w) off += wsum[w];
        if (tid < 256) gb[tid] = off + inc - v + table[(u64)tid * tiles + tile];
    }
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
    u32 k[LS_ROUNDS], p[LS_ROUNDS], rnk[LS_ROUNDS];
#pragma unroll
    for (int r = 0; r < LS_ROUNDS; ++r) {
        const u64 j = (u64)tile * LS_TILE + r * LS_THREADS + tid;
        const bool act = j < t;
        k[r] = act ? key[j] : 0u;
        p[r] = act ? (pay ? pay[j] : (u32)j) : 0u;
        const u32 d = (k[r] >> shift) & mask;
        u64 m = __ballot(act);  // the active lanes that share this lane's digit
        for (int b = 0; b < bits; ++b) {
            const bool one = (d >> b) & 1u;
            const u64 bal = __ballot(act && one);
            m &= one ? bal : ~bal;
        }
        rnk[r] = (u32)__popcll(m & lt);
        if (act && rnk[r] == 0) cc[r * (LS_THREADS / 64) + wid][d] = (unsigned short)__popcll(m);
    }
    __syncthreads();
    if (tid < 256) {  // exclusive prefix over the tile's chunks, in index order
        u32 run = 0;
#pragma unroll 8
        for (int c = 0; c < LS_CHUNKS; ++c) {
            const u32 v = cc[c][tid];
            cc[c][tid] = (unsigned short)run;
            run += v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LS_ROUNDS; ++r) {
        const u64 j = (u64)tile * LS_TILE + r * LS_THREADS + tid;
        if (j < t) {
            const u32 d = (k[r] >> shift) & mask;
            const u64 at = (u64)gb[d] + cc[r * (LS_THREADS / 64) + wid][d] + rnk[r];
            if (at < t) {  // (always: the histogram counted these same keys)
                if (key_out) key_out[at] = k[r];
                pay_out[at] = bad ? 0xFFFFFFFFu : p[r];
                if (last) x_sorted[at] = bad ? 0x7FF8000000000000ull : x[p[r]];
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_order_out(const u64 *__restrict__ words, const u64 *__restrict__ x, u64 t,
                                                   u32 *__restrict__ order, u64 *__restrict__ x_sorted,
                                                   const u32 *sort_err, const u32 *failed) {
    const bool bad = (*sort_err | *failed) != 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < t; i += (u64)gridDim.x * blockDim.x) {
        if (bad) {  // a sort gave up: no order at all rather than a wrong one
            order[i] = 0xFFFFFFFFu;
            x_sorted[i] = 0x7FF8000000000000ull;
            continue;
        }
        const u32 o = (u32)words[i];
        order[i] = o;
        x_sorted[i] = x[o];  // (the caller's own bits, signed zeros included)
    }
}

// estimate_i = pos[i] if the bisection counted a sample, else 1 (lib4mc/MonteCarloLib.py:53)
__device__ __forceinline__ double gn_estimate(const double *pos, const u64 *cnt, u64 i) {
    const double p = pos[i];
    return (cnt ? cnt[i] > 0 : !(p <= 0.0)) ? p : 1.0;  // (a NaN position stays NaN: never the estimate 1)
}

// ceil(est) - i * add as int64, or GN_NEG for an estimate that does not fit (first_bad keeps the lowest such index)
__device__ __forceinline__ i64 gn_term(const double *pos, const u64 *cnt, u64 i, int add1, u64 &first_bad) {
    const double c = ceil(gn_estimate(pos, cnt, i));
    if (!(c < 4611686018427387904.0)) {  // >= 2^62, or NaN
        first_bad = i < first_bad ? i : first_bad;
        return GN_NEG;
    }
    const i64 v = (i64)c - (add1 ? (i64)(i + 1) : 0);
    return v > 0 ? v : 0;  // (rank_0 = 0: the running maximum starts at 0)
}

__device__ __forceinline__ i64 block_max_scan_excl(i64 v, i64 *wave_tot, i64 &block_total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    i64 inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const i64 u = __shfl_up(inc, o);
        if (lane >= o && u > inc) inc = u;
    }
    i64 excl = __shfl_up(inc, 1);
    if (lane == 0) excl = GN_NEG;
    __syncthreads();
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    i64 off = GN_NEG, tot = GN_NEG;
    for (int w = 0; w < GN_THREADS / 64; ++w) {
        const i64 tt = wave_tot[w];
        if (w < wid && tt > off) off = tt;
        if (tt > tot) tot = tt;
    }
    block_total = tot;
    return off > excl ? off : excl;
}

// Two launches.  k_gn_chunk_max: per chunk, the largest term and the lowest index whose estimate does not fit int64.
// k_gn_ranks: every workgroup folds the chunks before it (a maximum of integers: exact in any order, so no scan kernel
// in between) and all chunks' lowest indices, then scans its own chunk; workgroup 0 writes the count of exact ranks.
__global__ void __launch_bounds__(GN_THREADS) k_gn_chunk_max(const double *__restrict__ pos, const u64 *__restrict__ cnt,
                                                             u64 t, u64 chunk, int add1, i64 *__restrict__ partials,
                                                             u64 *__restrict__ first_bad_of) {
    __shared__ i64 wave_tot[GN_THREADS / 64];
    __shared__ u64 wave_bad[GN_THREADS / 64];
    const u64 lo = (u64)blockIdx.x * chunk, hi = lo + chunk < t ? lo + chunk : t;
    i64 m = GN_NEG;
    u64 bad = ~0ull;
    for (u64 i = lo + threadIdx.x; i < hi; i += GN_THREADS) {
        const i64 v = gn_term(pos, cnt, i, add1, bad);
        m = v > m ? v : m;
    }
    i64 tot;
    (void)block_max_scan_excl(m, wave_tot, tot);
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const u64 u = __shfl_xor(bad, o);
        bad = u < bad ? u : bad;
    }
    if ((threadIdx.x & 63) == 0) wave_bad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < GN_THREADS / 64; ++w) bad = wave_bad[w] < bad ? wave_bad[w] : bad;
        partials[blockIdx.x] = tot;
        first_bad_of[blockIdx.x] = bad;
    }
}

// the queries of a quarter chunk answered (k_query<RUNS> of mc_rank.hip: same search, same outputs) and folded into the
// chunk maxima of k_gn_chunk_max; entry blockIdx.x of partials / first_bad_of
__global__ void __launch_bounds__(GN_THREADS) k_query_chunk_max(const double *__restrict__ vals, const u64 *__restrict__ ccum,
                                                                const double *__restrict__ wcum, const u64 *__restrict__ info,
                                                                const double *__restrict__ x, u64 t, u64 chunk, int add1,
                                                                double *__restrict__ out_pos, u64 *__restrict__ out_cnt,
                                                                i64 *__restrict__ partials, u64 *__restrict__ first_bad_of) {
    __shared__ i64 wave_tot[GN_THREADS / 64];
    __shared__ u64 wave_bad[GN_THREADS / 64];
    const int lane = threadIdx.x & 63;
    const u64 c0 = (u64)(blockIdx.x / GN_SUB) * chunk, c1 = c0 + chunk < t ? c0 + chunk : t;
    const u64 part = chunk / GN_SUB;  // (chunk is a multiple of GN_TILE: whole waves)
    const u64 lo_q = c0 + (u64)(blockIdx.x % GN_SUB) * part, hi_q = lo_q + part < c1 ? lo_q + part : c1;
    const u64 n = info[0];
    const bool failed = info[1] != 0;  // a failed build must not pass for a result
    i64 m = GN_NEG;
    u64 bad = ~0ull;
    for (u64 q0 = lo_q + (threadIdx.x - lane); q0 < hi_q; q0 += GN_THREADS) {
        const u64 q = q0 + lane;
        const bool act = q < hi_q;
        const double v = act ? x[q] : 0.0;
        double vmin = act ? v : __longlong_as_double(0x7FF0000000000000ll);
        double vmax = act ? v : __longlong_as_double(0xFFF0000000000000ull);
        const bool odd = act && !(v == v);
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const double a2 = __shfl_xor(vmin, o), b2 = __shfl_xor(vmax, o);
            vmin = a2 < vmin ? a2 : vmin;
            vmax = b2 > vmax ? b2 : vmax;
        }
        u64 lo = 0, hi = failed ? 0 : n;
        if (!failed && __ballot(odd) == 0 && n > 4096) {
            const double w = lane < 32 ? vmin : vmax;
            const int sub = lane & 31;
            u64 a = 0, b = n;
            while (b - a > 64) {
                const u64 step = (b - a) / 33 + 1;
                const u64 p = a + step * (u64)(sub + 1) - 1;
                const bool gt = p < b ? vals[p] > w : true;
                const u64 bal = __ballot(gt);
                const u32 half = lane < 32 ? (u32)bal : (u32)(bal >> 32);
                const int first = half ? __ffs((int)half) - 1 : 32;
                const u64 nb2 = first < 32 ? a + step * (u64)(first + 1) - 1 : b;
                const u64 na = first > 0 ? a + step * (u64)first : a;
                a = na;
                b = nb2 < b ? nb2 : b;
            }
            lo = __shfl(a, 0);
            hi = __shfl(b, 32);
        }
        if (act) {
            while (lo < hi) {  // first index with vals[i] > v  (bisect_right)
                const u64 mid = (lo + hi) >> 1;
                if (vals[mid] > v) hi = mid; else lo = mid + 1;
            }
            const u64 c = failed ? ~0ull : (lo ? ccum[lo - 1] : 0ull);  // (as k_query<RUNS>: never an estimate of 1)
            const double p = failed ? __longlong_as_double(0x7FF8000000000000ll) : (lo ? wcum[lo - 1] : 0.0);
            out_cnt[q] = c;
            out_pos[q] = p;
            // (gn_term on the values just written)
            const double ce = ceil(c > 0 ? p : 1.0);
            i64 term;
            if (!(ce < 4611686018427387904.0)) {
                bad = q < bad ? q : bad;
                term = GN_NEG;
            } else {
                term = (i64)ce - (add1 ? (i64)(q + 1) : 0);
                term = term > 0 ? term : 0;
            }
            m = term > m ? term : m;
        }
    }
    i64 tot;
    (void)block_max_scan_excl(m, wave_tot, tot);
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const u64 u = __shfl_xor(bad, o);
        bad = u < bad ? u : bad;
    }
    if (lane == 0) wave_bad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < GN_THREADS / 64; ++w) bad = wave_bad[w] < bad ? wave_bad[w] : bad;
        partials[blockIdx.x] = tot;
        first_bad_of[blockIdx.x] = bad;
    }
}

__global__ void __launch_bounds__(GN_THREADS) k_gn_ranks(const double *__restrict__ pos, const u64 *__restrict__ cnt,
                                                         u64 t, u64 chunk, int add1, const i64 *__restrict__ partials,
                                                         const u64 *__restrict__ first_bad_of, u32 nb, u64 *out_exact,
                                                         i64 *__restrict__ rank, u32 sub) {
    __shared__ i64 wave_tot[GN_THREADS / 64];
    __shared__ u64 wave_bad[GN_THREADS / 64];
    __shared__ i64 s_carry;
    __shared__ u64 s_exact;
    const u64 lo = (u64)blockIdx.x * chunk, hi = lo + chunk < t ? lo + chunk : t;
    {
        i64 m = GN_NEG;
        u64 bad = t;  // (no estimate out of range: all t ranks are exact)
        for (u32 i = threadIdx.x; i < nb * sub; i += GN_THREADS) {  // (`sub` entries a chunk: k_query_chunk_max's workgroups)
            if (i < blockIdx.x * sub) m = partials[i] > m ? partials[i] : m;
            bad = first_bad_of[i] < bad ? first_bad_of[i] : bad;
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const i64 u = __shfl_xor(m, o);
            m = u > m ? u : m;
            const u64 b2 = __shfl_xor(bad, o);
            bad = b2 < bad ? b2 : bad;
        }
        if ((threadIdx.x & 63) == 0) {
            wave_tot[threadIdx.x >> 6] = m;
            wave_bad[threadIdx.x >> 6] = bad;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < GN_THREADS / 64; ++w) {
                m = wave_tot[w] > m ? wave_tot[w] : m;
                bad = wave_bad[w] < bad ? wave_bad[w] : bad;
            }
            s_carry = m;
            s_exact = bad;
            if (blockIdx.x == 0) *out_exact = bad;
        }
        __syncthreads();
    }
    const u64 exact = s_exact;
    i64 carry = s_carry;
    if (carry < 0) carry = 0;
    u64 unused = ~0ull;
    for (u64 t0 = lo; t0 < hi; t0 += GN_TILE) {
        // a lane scans GN_ITEMS consecutive terms
        i64 v[GN_ITEMS];
        i64 m = GN_NEG;
#pragma unroll
        for (int j = 0; j < GN_ITEMS; ++j) {
            const u64 i = t0 + (u64)threadIdx.x * GN_ITEMS + j;
            v[j] = i < hi ? gn_term(pos, cnt, i, add1, unused) : GN_NEG;
            m = v[j] > m ? v[j] : m;
            v[j] = m;  // inclusive within the lane
        }
        i64 tot;
        const i64 excl = block_max_scan_excl(m, wave_tot, tot);
        const i64 before = excl > carry ? excl : carry;
#pragma unroll
        for (int j = 0; j < GN_ITEMS; ++j) {
            const u64 i = t0 + (u64)threadIdx.x * GN_ITEMS + j;
            if (i < hi) {
                const i64 r = v[j] > before ? v[j] : before;
                // from the first estimate beyond int64 on: the estimates themselves, for the caller's integers
                rank[i] = i < exact ? r + (add1 ? (i64)(i + 1) : 0) : (i64)__double_as_longlong(gn_estimate(pos, cnt, i));
            }
        }
        carry = tot > carry ? tot : carry;
        __syncthreads();
    }
}

}  // namespace

// digit tables of the LSD passes (t <= LS_MAX_T): 256 counters a tile + 256 totals, in a region of their own behind
// the failure word -- the head of the sort's workspace holds its control block and the hint a later sort reuses
static size_t lsd_table_bytes(uint64_t t) {
    if (t > LS_MAX_T) return 0;
    return align256(((size_t)256 * ((t + LS_TILE - 1) / LS_TILE) + 256) * sizeof(u32));
}

extern "C" size_t mc_order_workspace_bytes(uint64_t t) {
    return align256(mc_sort_workspace_bytes(t)) + 2 * align256(t * sizeof(u64)) + 256 + lsd_table_bytes(t);
}

namespace {

struct OrderPlan {
    u64 *words, *words_sorted;
    u32 *failed;
    size_t sort_bytes;
    u32 g, gw, stride;
};

// the LSD passes over the index L < t: how many, and how many bits each
void lsd_passes(uint64_t t, int &passes, int &width) {
    int nbits = 1;
    while (nbits < 32 && (t - 1) >> nbits) ++nbits;
    passes = (nbits + 7) / 8;
    width = (nbits + passes - 1) / passes;
}

OrderPlan order_plan(uint64_t t, void *workspace) {
    OrderPlan o;
    o.sort_bytes = align256(mc_sort_workspace_bytes(t));
    o.words = (u64 *)((char *)workspace + o.sort_bytes);
    o.words_sorted = (u64 *)((char *)o.words + align256(t * sizeof(u64)));
    o.failed = (u32 *)((char *)o.words_sorted + align256(t * sizeof(u64)));
    o.g = (u32)((t + 255) / 256 < 2048 ? (t + 255) / 256 : 2048);
    o.stride = (u32)((t + OW_LDS - 1) / OW_LDS) ? (u32)((t + OW_LDS - 1) / OW_LDS) : 1u;  // (t < 2^32)
    o.gw = (u32)((t + 2047) / 2048 < 512 ? (t + 2047) / 2048 : 512);
    return o;
}

int order_args(const char *who, const double *x, uint64_t t, double *x_sorted, void *workspace, size_t workspace_bytes) {
    if (!x || !x_sorted || !workspace || (const void *)x == (const void *)x_sorted) {
        mc_set_error("%s: bad argument", who);
        return MC_ERR_ARG;
    }
    if (t >= 0xFFFFFFFFull) {
        mc_set_error("%s: at most 2^32 - 2 keys (got %llu)", who, (unsigned long long)t);
        return MC_ERR_ARG;
    }
    if (workspace_bytes < mc_order_workspace_bytes(t)) {
        mc_set_error("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, mc_order_workspace_bytes(t));
        return MC_ERR_WORKSPACE;
    }
    return MC_OK;
}

}  // namespace

// The first half of mc_order_keys: the key sort and, per password, the index of the first sorted key equal to its
// own.  x_values (or null) receives the scores in ascending order AS NUMBERS (both zeros as +0.0): all that the
// bisection needs, so a caller may put its queries while mc_order_keys_end is still finishing the order of the
// passwords on another stream.  x_sorted holds the sort's order words until mc_order_keys_end has run.
extern "C" int mc_order_keys_begin(const double *x, uint64_t t, double *x_sorted, double *x_values, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (t == 0) return MC_OK;
    int rc = order_args("mc_order_keys_begin", x, t, x_sorted, workspace, workspace_bytes);
    if (rc != MC_OK) return rc;
    if ((const void *)x_values == (const void *)x || (const void *)x_values == (const void *)x_sorted) {
        mc_set_error("mc_order_keys_begin: x_values must be a buffer of its own");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const OrderPlan o = order_plan(t, workspace);
    uint32_t *err1 = nullptr;
    // (the sort reads the scores as they are and maps them to order words on the way: no pass of our own for that)
    rc = mc_sort_keys((const uint64_t *)x, t, (uint64_t *)x_sorted, workspace, o.sort_bytes, &err1, MC_SORT_RAW_F64, s);
    if (rc != MC_OK) return rc;
    const u32 tiles = (u32)((t + LS_TILE - 1) / LS_TILE);
    if (t <= LS_MAX_T) {
        // (the search walks the LSD passes' tiles and leaves the first pass's digit histograms in their table)
        int passes, width;
        lsd_passes(t, passes, width);
        u32 *table = (u32 *)((char *)o.failed + 256);
        // (a workgroup walks whole tiles: none beyond their number -- each stages its keys first, for nothing if it has no tile)
        hipLaunchKernelGGL(k_order_words<false>, dim3(o.gw < tiles ? o.gw : tiles), dim3(OW_THREADS), 0, s, (const u64 *)x_sorted, (const u64 *)x,
                           (u64)t, o.stride, o.words, (const u32 *)err1, o.failed, (u64 *)x_values, table, tiles,
                           (1u << width) - 1u);
    } else {
        hipLaunchKernelGGL(k_order_words<true>, dim3(o.gw), dim3(OW_THREADS), 0, s, (const u64 *)x_sorted, (const u64 *)x,
                           (u64)t, o.stride, o.words, (const u32 *)err1, o.failed, (u64 *)x_values, (u32 *)nullptr, tiles, 0u);
    }
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

// The second half: the passwords' stable order from what mc_order_keys_begin left in the workspace (same x, t,
// x_sorted and workspace), and x_sorted as the caller's own bits.
extern "C" int mc_order_keys_end(const double *x, uint64_t t, double *x_sorted, uint32_t *order, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (t == 0) return MC_OK;
    int rc = order_args("mc_order_keys_end", x, t, x_sorted, workspace, workspace_bytes);
    if (rc != MC_OK) return rc;
    if (!order) {
        mc_set_error("mc_order_keys_end: bad argument");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const OrderPlan o = order_plan(t, workspace);
    if (t <= LS_MAX_T) {
        // (L, index) pairs through stable LSD passes; the buffers of the words hold two u32 arrays each, the
        // digit tables have a region of their own (lsd_table_bytes)
        u32 *ka = (u32 *)o.words, *pa = ka + t, *kb = (u32 *)o.words_sorted, *pb = kb + t;
        const u32 tiles = (u32)((t + LS_TILE - 1) / LS_TILE);
        u32 *table = (u32 *)((char *)o.failed + 256), *totals = table + (size_t)256 * tiles;
        int passes, width;
        lsd_passes(t, passes, width);
        const u32 *kin = ka, *pin = nullptr;  // (first pass: the index is the position)
        for (int ps = 0; ps < passes; ++ps) {
            const bool last = ps + 1 == passes;  // (writes the caller's order and sorted scores itself)
            u32 *kout = (ps & 1) ? ka : kb, *pout = last ? order : ((ps & 1) ? pa : pb);
            const u32 shift = (u32)(ps * width), mask = (1u << width) - 1u;
            if (ps)  // (the first pass's histograms are mc_order_keys_begin's: its search kernel counted them)
                hipLaunchKernelGGL(k_lsd_hist, dim3(tiles), dim3(LS_THREADS), 0, s, kin, (u64)t, shift, mask, tiles, table);
            hipLaunchKernelGGL(k_lsd_scan, dim3(mask + 1u), dim3(256), 0, s, table, tiles, totals);  // (a workgroup a digit in use)
            hipLaunchKernelGGL(k_lsd_scatter, dim3(tiles), dim3(LS_THREADS), 0, s, kin, pin, (u64)t, shift, mask, width,
                               tiles, (const u32 *)table, (const u32 *)totals, last ? (u32 *)nullptr : kout, pout,
                               last ? (const u64 *)x : (const u64 *)nullptr, (u64 *)x_sorted, (const u32 *)o.failed);
            kin = kout;
            pin = pout;
        }
        HIP_TRY(hipGetLastError());
        return MC_OK;
    }
    uint32_t *err2 = nullptr;
    rc = mc_sort_keys((const uint64_t *)o.words, t, (uint64_t *)o.words_sorted, workspace, o.sort_bytes, &err2,
                      MC_SORT_FEW_REPEATS /* the words are distinct */, s);
    if (rc != MC_OK) return rc;
    hipLaunchKernelGGL(k_order_out, dim3(o.g), dim3(256), 0, s, (const u64 *)o.words_sorted, (const u64 *)x, (u64)t, order,
                       (u64 *)x_sorted, (const u32 *)err2, (const u32 *)o.failed);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_order_keys(const double *x, uint64_t t, double *x_sorted, uint32_t *order, void *workspace,
                             size_t workspace_bytes, void *stream) {
    if (t == 0) return MC_OK;
    if (!order) {
        mc_set_error("mc_order_keys: bad argument");
        return MC_ERR_ARG;
    }
    const int rc = mc_order_keys_begin(x, t, x_sorted, nullptr, workspace, workspace_bytes, stream);
    if (rc != MC_OK) return rc;
    return mc_order_keys_end(x, t, x_sorted, order, workspace, workspace_bytes, stream);
}

extern "C" size_t mc_guess_numbers_workspace_bytes(uint64_t t) {
    (void)t;
    return 2 * align256(GN_MAX_BLOCKS * GN_SUB * sizeof(i64)) + 256;  // per chunk (and workgroup of a chunk): largest term, lowest index out of range
}

extern "C" int mc_guess_numbers(const double *pos, const uint64_t *cnt, uint64_t t, int add1, int64_t *out_rank,
                                uint64_t *out_exact, void *workspace, size_t workspace_bytes, void *stream) {
    if (!out_exact) {
        mc_set_error("mc_guess_numbers: bad argument");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (t == 0) {
        HIP_TRY(hipMemsetAsync(out_exact, 0, sizeof(uint64_t), s));
        return MC_OK;
    }
    if (!pos || !out_rank || !workspace) {
        mc_set_error("mc_guess_numbers: bad argument");
        return MC_ERR_ARG;
    }
    if (workspace_bytes < mc_guess_numbers_workspace_bytes(t)) {
        mc_set_error("mc_guess_numbers: workspace of %zu bytes, need %zu", workspace_bytes,
                     mc_guess_numbers_workspace_bytes(t));
        return MC_ERR_WORKSPACE;
    }
    i64 *partials = (i64 *)workspace;
    const u64 tiles = (t + GN_TILE - 1) / GN_TILE;
    const u32 nb = (u32)(tiles < (u64)GN_MAX_BLOCKS ? tiles : (u64)GN_MAX_BLOCKS);
    const u64 chunk = ((tiles + nb - 1) / nb) * GN_TILE;
    u64 *first_bad_of = (u64 *)((char *)workspace + align256(GN_MAX_BLOCKS * GN_SUB * sizeof(i64)));
    hipLaunchKernelGGL(k_gn_chunk_max, dim3(nb), dim3(GN_THREADS), 0, s, pos, (const u64 *)cnt, (u64)t, chunk, add1,
                       partials, first_bad_of);
    hipLaunchKernelGGL(k_gn_ranks, dim3(nb), dim3(GN_THREADS), 0, s, pos, (const u64 *)cnt, (u64)t, chunk, add1,
                       (const i64 *)partials, (const u64 *)first_bad_of, nb, (u64 *)out_exact, (i64 *)out_rank, 1u);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

// ml2p_iter2gc as it stands in the reference -- the bisection INSIDE the loop over the passwords
// (lib4mc/MonteCarloLib.py:47-55) -- on the run index: mc_rank_runs_query and mc_guess_numbers in two launches instead
// of three.  The first kernel answers the queries (k_query's search: a wave brackets its 64 ascending queries in rounds
// of 32 probes, then every lane bisects), writes pos / cnt as mc_rank_runs_query does and folds the chunk maxima of
// mc_guess_numbers' first kernel on the way; the second is k_gn_ranks as it is.  Same outputs bit for bit.
extern "C" int mc_rank_runs_guess_numbers(const double *vals, const uint64_t *cum_count, const double *cum_pos,
                                          const uint64_t *info, const double *x, uint64_t t, int add1, double *out_pos,
                                          uint64_t *out_cnt, int64_t *out_rank, uint64_t *out_exact, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    if (!out_exact) {
        mc_set_error("mc_rank_runs_guess_numbers: bad argument");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (t == 0) {
        HIP_TRY(hipMemsetAsync(out_exact, 0, sizeof(uint64_t), s));
        return MC_OK;
    }
    if (!vals || !cum_count || !cum_pos || !info || !x || !out_pos || !out_cnt || !out_rank || !workspace) {
        mc_set_error("mc_rank_runs_guess_numbers: bad argument");
        return MC_ERR_ARG;
    }
    if (workspace_bytes < mc_guess_numbers_workspace_bytes(t)) {
        mc_set_error("mc_rank_runs_guess_numbers: workspace of %zu bytes, need %zu", workspace_bytes,
                     mc_guess_numbers_workspace_bytes(t));
        return MC_ERR_WORKSPACE;
    }
    i64 *partials = (i64 *)workspace;
    const u64 tiles = (t + GN_TILE - 1) / GN_TILE;
    const u32 nb = (u32)(tiles < (u64)GN_MAX_BLOCKS ? tiles : (u64)GN_MAX_BLOCKS);
    const u64 chunk = ((tiles + nb - 1) / nb) * GN_TILE;
    u64 *first_bad_of = (u64 *)((char *)workspace + align256(GN_MAX_BLOCKS * GN_SUB * sizeof(i64)));
    hipLaunchKernelGGL(k_query_chunk_max, dim3(nb * GN_SUB), dim3(GN_THREADS), 0, s, vals, (const u64 *)cum_count, cum_pos,
                       (const u64 *)info, x, (u64)t, chunk, add1, out_pos, (u64 *)out_cnt, partials, first_bad_of);
    hipLaunchKernelGGL(k_gn_ranks, dim3(nb), dim3(GN_THREADS), 0, s, (const double *)out_pos, (const u64 *)out_cnt, (u64)t,
                       chunk, add1, (const i64 *)partials, (const u64 *)first_bad_of, nb, (u64 *)out_exact,
                       (i64 *)out_rank, (u32)GN_SUB);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}
