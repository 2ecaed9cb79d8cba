// Rank estimation kernels (gfx950): replaces MonteCarloLib.__init__ / ml2p2rank,
// reference lib4mc/MonteCarloLib.py:10-32:
//     keys.sort(); positions = cumsum(2 ** (keys - log2 n)); bisect_right(keys, x)
//
//   sort      onesweep LSD radix sort of the fp64 keys as u64 (keys are >= +0.0, so IEEE bit
//             order is numeric order; NaN = failed sample sorts last), 8-bit digits.  One
//             upfront kernel builds all 8 digit histograms; passes whose digit is constant over
//             the whole input (the sign/exponent bytes, typically) are skipped and the
//             ping-pong is planned so the last pass lands in keys_sorted.  Each active pass is
//             ONE kernel that reads every key once and writes it once: a (persistent) workgroup
//             takes its next tile, ranks its 6144 keys by digit (wave ballots), publishes the
//             tile's digit counts, stages the keys in LDS in digit order while its first
//             look-back reads travel, obtains the counts of all earlier tiles by decoupled
//             look-back over the published words, and writes the staged keys out so that
//             global writes are digit-contiguous runs.
//   scan      positions = inclusive prefix sum of exp2(key - log2 n_global), fp64,
//             reduce-then-scan over per-workgroup chunks.
//   query     one lane per test password: upper_bound over the sorted keys.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>

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

constexpr int SORT_TILE_MIN = 4096;  // smallest tile any variant uses: sizes the look-back words
// look-back word of (tile, digit): 2 flag bits + count
constexpr unsigned long long LB_LOCAL = 1ull << 62;  // count of this tile alone
constexpr unsigned long long LB_INCL = 2ull << 62;   // count of this tile and every earlier one
constexpr unsigned long long LB_MASK = (1ull << 62) - 1;
constexpr unsigned int LB_SPIN_LIMIT = 1u << 23;     // a bound, not a protocol: see k_onesweep
constexpr int LB_WINDOW = 4;                         // predecessors read per look-back step
constexpr int HIST_COPIES = 4, HIST_PITCH = 8 * 256 + 1;
constexpr size_t HIST_WORDS = 8 * 256 + 8 + 8;  // histograms, spare, error word (+pad)
constexpr int HIST_SPREAD = 4;                  // copies of the bins the workgroups of k_hist_all add into
constexpr size_t HIST_ALL_WORDS = HIST_WORDS + (size_t)(HIST_SPREAD - 1) * 8 * 256;
constexpr int HIST_PLAIN = 5;  // digits counted without looking for wave-wide agreement
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;  // (k_positions pads its LDS tile for exactly 8)
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr int SCAN_MAX_BLOCKS = 2048;

// workspace layout (u32 words unless noted)
struct Plan {
    u32 skip[8];   // pass is a no-op
    u32 src[8];    // 0 = keys_in, 1 = keys_sorted (A), 2 = temp (B)
    u32 dst[8];
    u32 slot[8];   // look-back region (0/1) the pass uses; it clears the other one for the next pass
    u32 n_active;
};

struct SortWs {
    u64 *temp;        // [n] second key buffer
    u32 *hist;        // [8][256], 8 spare words, then the error flag
    u64 *lookback;    // [2][tiles][256]
    Plan *plan;
    double *partials; // [scan blocks]
};

__device__ __forceinline__ u32 digit_of(u64 k, int pass) { return (u32)(k >> (8 * pass)) & 255u; }

// ---- all 8 digit histograms in one read ------------------------------------------------------
__global__ void __launch_bounds__(256) k_hist_all(const u64 *__restrict__ keys, u64 n, u32 *__restrict__ hist) {
    // HIST_COPIES copies of the 8 x 256 bins, lane l adds to copy l % HIST_COPIES: sampled keys repeat
    // (one password drawn many times, few exponent values), and lanes adding to one LDS word serialise.
    // Copies are HIST_PITCH words apart so that one bin's copies fall into different banks.
    __shared__ u32 hc[HIST_COPIES * HIST_PITCH];
    for (int i = threadIdx.x; i < HIST_COPIES * HIST_PITCH; i += blockDim.x) hc[i] = 0;
    u32 *h = hc + (threadIdx.x % HIST_COPIES) * HIST_PITCH;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    // four keys per lane and round, their loads issued together; whole waves stay in the loop so the
    // ballots see every lane (a lane past the end holds no key)
    for (u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x; i0 - threadIdx.x % 64 < n; i0 += 4 * stride) {
        u64 k[4];
        bool have[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u64 i = i0 + r * stride;
            have[r] = i < n;
            k[r] = have[r] ? keys[i] : 0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u64 mask = __ballot(have[r]);
            if (mask == 0) continue;
            const u64 first = __shfl(k[r], __ffsll((long long)mask) - 1);
            const u64 diff = have[r] ? (k[r] ^ first) : 0;
            const u32 lanes = __popcll(mask);
            const bool leader = have[r] && __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0;
            // the low bytes are mantissa bits: lanes agree on one only when they hold the same key (a
            // run of one repeated password), which is the one case worth looking for
            const bool one_key = __ballot(diff != 0) == 0;
#pragma unroll
            for (int p = 0; p < HIST_PLAIN; ++p) {
                if (one_key) {
                    if (leader) atomicAdd(&h[p * 256 + digit_of(k[r], p)], lanes);
                } else if (have[r]) {
                    atomicAdd(&h[p * 256 + digit_of(k[r], p)], 1u);
                }
            }
#pragma unroll
            for (int p = HIST_PLAIN; p < 8; ++p) {
                // a digit the whole wave agrees on (sign/exponent bytes; runs of one repeated password)
                // costs one LDS add, not 64 colliding ones
                const bool same = __ballot(((diff >> (8 * p)) & 255ull) != 0) == 0;
                if (same) {
                    if (leader) atomicAdd(&h[p * 256 + digit_of(k[r], p)], lanes);
                } else if (have[r]) {
                    atomicAdd(&h[p * 256 + digit_of(k[r], p)], 1u);
                }
            }
        }
    }
    __syncthreads();
    // (few workgroups: every global bin takes one atomic per workgroup, and atomics on one address
    // serialise at ~11 ns each; the workgroups finish together, so they add into HIST_SPREAD copies
    // of the bins, which k_plan sums: 0.546 -> 0.532 ms per build of 1e7 keys, the same for 2 to 16 copies)
    const u32 copy = blockIdx.x % HIST_SPREAD;
    u32 *mine_g = copy ? hist + HIST_WORDS + (size_t)(copy - 1) * 8 * 256 : hist;
    for (int i = threadIdx.x; i < 8 * 256; i += blockDim.x) {
        u32 v = 0;
        for (int c = 0; c < HIST_COPIES; ++c) v += hc[c * HIST_PITCH + i];
        if (v) atomicAdd(&mine_g[i], v);
    }
}

// one workgroup: which passes do anything, and which buffer each reads/writes
__global__ void k_plan(u32 *hist, u64 n, Plan *plan) {
    __shared__ u32 trivial[8];
    if (threadIdx.x < 8) trivial[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 256; i += blockDim.x) {
        u32 v = hist[i];
        for (int c = 1; c < HIST_SPREAD; ++c) v += hist[HIST_WORDS + (size_t)(c - 1) * 8 * 256 + i];
        hist[i] = v;  // the passes read the totals here
        if ((u64)v == n) trivial[i >> 8] = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 active = 0;
        for (int p = 0; p < 8; ++p) active += trivial[p] ? 0 : 1;
        // destinations alternate and the LAST active pass must write A (keys_sorted)
        u32 cur_src = 0, k = 0;
        for (int p = 0; p < 8; ++p) {
            plan->skip[p] = trivial[p];
            plan->src[p] = plan->dst[p] = plan->slot[p] = 0;
            if (trivial[p]) continue;
            u32 remaining_after = active - k - 1;
            u32 d = (remaining_after % 2 == 0) ? 1 : 2;
            plan->src[p] = cur_src;
            plan->dst[p] = d;
            plan->slot[p] = k & 1u;
            cur_src = d;
            ++k;
        }
        plan->n_active = active;
    }
}

__device__ __forceinline__ const u64 *pick_src(u32 id, const u64 *in, const u64 *a, const u64 *b) {
    return id == 0 ? in : (id == 1 ? a : b);
}

// nothing to sort by (all passes trivial): keys_sorted = keys_in
__global__ void __launch_bounds__(256) k_copy_if_unsorted(const Plan *__restrict__ plan, const u64 *__restrict__ in,
                                                          u64 *__restrict__ out, u64 n) {
    if (plan->n_active != 0) return;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}

// ---- one radix pass, one kernel (onesweep) -------------------------------------------------------
// Persistent workgroups: the grid is no larger than what the GPU holds at once (the host sizes it
// from the kernel's occupancy), and workgroup g takes tiles g, g + G, g + 2G, ...  A tile only ever
// waits for LOWER tiles; the lowest unfinished tile belongs to a workgroup that is resident or will
// be made resident without anyone else finishing, and it waits for nothing: the look-back cannot
// deadlock whatever order the hardware dispatches in.  (A ticket counter would give the same
// guarantee, but 2,441 atomics on one address cost 13 us per pass: same-address atomics serialise
// at ~11 ns each.)  That argument needs the GPU to itself: when ANOTHER process holds part of the GPU
// with a grid of the same kind (two ranks rehearsing on one GPU), each can be left with half a grid
// resident that waits for the half that cannot start.  TICKETS is the variant for a shared GPU
// (MC_SORT_TICKETS=1): tiles are taken from a counter in increasing order, two ahead so that nobody
// waits for the counter, and the lowest unfinished tile is then always held by a running workgroup
// whatever is resident; measured +21 us per pass at 1e7 keys, which is why it is not the default.
// The spin is nevertheless bounded; if the bound is ever hit the error word is
// set and k_positions poisons the output with NaN (loud, not a hang).  Look-back words are written
// and read with agent-scope atomics (the per-XCD L2s are not coherent with each other); nothing
// else crosses workgroups.
template <int THREADS, int ITEMS, int PREFETCH, int WAVES, bool TICKETS = false>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))
k_onesweep(const Plan *__restrict__ plan, int pass, const u64 *in, u64 *a, u64 *b, u64 n, u32 tiles,
           const u32 *__restrict__ hist, u32 *err, u64 *lookback) {
    constexpr int TILE = THREADS * ITEMS, NW = THREADS / 64;
    if (plan->skip[pass]) return;
    // an earlier pass gave up (spin bound, count mismatch): the result is poisoned anyway, do not spin again
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const u64 *src = pick_src(plan->src[pass], in, a, b);
    u64 *dst = plan->dst[pass] == 1 ? a : b;
    u64 *lb = lookback + (u64)plan->slot[pass] * tiles * 256;
    u64 *lb_next = lookback + (u64)(1u - plan->slot[pass]) * tiles * 256;
    __shared__ u64 stage[TILE];
    __shared__ u32 wcnt[NW][256];  // per-wave digit counts of the tile, then exclusive prefix over waves
    __shared__ u64 base[256];      // global write offset of each digit for this tile
    __shared__ u32 tile_off[256];  // exclusive scan of the tile's digit counts
    __shared__ u32 wave_sum[4];
    __shared__ u32 wave_below[4];
    __shared__ u32 ticket_s[1];
    u32 *ticket = err - 8 + pass;  // (TICKETS) the pass's counter among the spare words before the error word
    // each wave owns 64 * ITEMS consecutive keys of the tile: order = (wave, round, lane).  The 256
    // digits belong to the first 256 lanes (waves 0..3).
    u64 key[ITEMS], nkey[PREFETCH ? ITEMS : 1];
    auto load_tile = [&](u32 tile, u64 *k) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const u64 t0 = (u64)tile * TILE;
        const u32 tile_n = tile < tiles ? (u32)min((u64)TILE, n - t0) : 0u;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const u32 j = wid * (64 * ITEMS) + r * 64 + lane;
            k[r] = j < tile_n ? src[t0 + j] : ~0ull;
        }
    };
    u32 tile = blockIdx.x, tile1 = blockIdx.x + gridDim.x, ahead = 0;
    if (TICKETS) {
        if (threadIdx.x == 0) ticket_s[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        tile = tile1 = ticket_s[0];  // (tile1 only matters to the prefetching forms, which take no tickets)
    }
    if (PREFETCH) load_tile(tile, nkey);
    while (tile < tiles) {
    // the lane's identity is made opaque once per tile: what the compiler derives from it (shuffle lane
    // indices, lane masks, LDS addresses) is then recomputed per tile instead of being hoisted out of
    // the persistent loop and held in ~80 VGPRs for the whole kernel (256 x 24 needed > 256 VGPRs
    // against 175 for the same code without the loop, which is what held it at two waves per SIMD)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < NW * 256; i += THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();  // (also: every wave has left the previous tile's write-out and read ticket_s)
    const u64 t0 = (u64)tile * TILE;
    const u32 tile_n = (u32)min((u64)TILE, n - t0);
    if (tid < 256) lb_next[(u64)tile * 256 + tid] = 0;  // the next active pass starts from clean words

    u32 rank[ITEMS];
    if (PREFETCH) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) key[r] = nkey[r];
    } else {
        load_tile(tile, key);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const u32 j = wid * (64 * ITEMS) + r * 64 + lane;
        const bool valid = j < tile_n;
        const u32 d = digit_of(key[r], pass);
        // lanes of this wave holding the same digit (invalid lanes match nobody)
        u64 peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const u64 vote = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? vote : ~vote;
        }
        const u32 before = __popcll(peers & ((1ull << lane) - 1ull));
        const u32 cnt = __popcll(peers);
        const int leader = __ffsll((long long)peers) - 1;
        u32 old = 0;
        if (valid && lane == leader) old = atomicAdd(&wcnt[wid][d], cnt);
        old = __shfl(old, leader < 0 ? 0 : leader);
        rank[r] = old + before;
    }
    // the next tile, read at the end of this one; drawn by the LAST wave, which has no look-back to
    // run: when every workgroup of a round asks at once the answer takes microseconds (same-address
    // atomics queue at ~11 ns each), and in wave 0 the look-back's loads would wait behind it
    if (TICKETS && tid == THREADS - 1) ahead = atomicAdd(ticket, 1u);
    // the next tile's keys travel while this tile waits for its predecessors and writes out
    if (PREFETCH == 1) load_tile(tile1, nkey);
    __syncthreads();
    u32 tot = 0, inc = 0, ginc = 0, gown = 0, gbelow = 0;
    u64 *mine = lb + (u64)tile * 256 + (tid & 255);
    if (tid < 256) {
        // digit `tid`: exclusive prefix over the waves, tile total
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const u32 c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
        tot = run;
        __hip_atomic_store(mine, (u64)tot | (tile == 0 ? LB_INCL : LB_LOCAL), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        // exclusive scans over the 256 digits: of the tile's counts and of the global histogram
        inc = tot;
        ginc = gown = hist[pass * 256 + tid];
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = __shfl_up(inc, o), g = __shfl_up(ginc, o);
            if (lane >= o) {
                inc += v;
                ginc += g;
            }
        }
        if (lane == 63) {
            wave_sum[wid] = inc;
            wave_below[wid] = ginc;
        }
    }
    __syncthreads();
    if (tid < 256) {
        u32 woff = 0;
        for (int w = 0; w < wid; ++w) {
            woff += wave_sum[w];
            gbelow += wave_below[w];
        }
        tile_off[tid] = woff + inc - tot;
        gbelow += ginc - gown;  // keys of lower digits in the whole input
    }
    __syncthreads();
    // the first look-back reads travel while the keys are staged
    u64 v[LB_WINDOW];
    bool have = false;
    if (tid < 256 && tile > 0) {
#pragma unroll
        for (int j = 0; j < LB_WINDOW; ++j)
            v[j] = (long long)tile - 1 - j >= 0
                       ? __hip_atomic_load(lb + (u64)(tile - 1 - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : LB_INCL;
        have = true;
    }
    // stage keys in LDS in digit order BEFORE asking for the predecessors' counts: it needs nothing from
    // them, and they get that much longer to publish (the look-back wait is the largest single cost of
    // a pass, ~17 of 62 us); key[] and rank[] are dead from here on
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const u32 j = wid * (64 * ITEMS) + r * 64 + lane;
        if (j < tile_n) {
            const u32 d = digit_of(key[r], pass);
            const u32 at = tile_off[d] + wcnt[wid][d] + rank[r];
            if (at < (u32)TILE) stage[at] = key[r]; else atomicOr(err, 2u);
        }
    }
    if (tid < 256) {
        // decoupled look-back: keys of digit `tid` in all earlier tiles, LB_WINDOW predecessors per
        // round trip (the reads are uncached; windows of 1..8 perform alike, 32 is slower).
        u64 excl = 0;
        if (tile > 0) {
            u32 spins = 0;
            long long p = (long long)tile - 1;  // nearest predecessor not yet accounted for
            bool done = false;
            while (!done) {
                if (!have) {
#pragma unroll
                    for (int j = 0; j < LB_WINDOW; ++j)
                        v[j] = p - j >= 0 ? __hip_atomic_load(lb + (u64)(p - j) * 256 + tid, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT)
                                          : LB_INCL;
                }
                have = false;
                int used = 0;
#pragma unroll
                for (int j = 0; j < LB_WINDOW; ++j) {
                    if (done || used != j) continue;
                    const u64 f = v[j] & ~LB_MASK;
                    if (f == 0) continue;  // not published yet: poll again from here
                    excl += v[j] & LB_MASK;
                    used = j + 1;
                    if (f != LB_LOCAL) done = true;
                }
                p -= used;
                if (!done && used < LB_WINDOW) {
                    if (++spins > LB_SPIN_LIMIT) {
                        atomicOr(err, 1u);
                        done = true;
                    }
                    // somebody else already gave up: every later wait would only run to the bound as well
                    if ((spins & 63u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        done = true;
                    // back off: a tile that has waited long is waiting for a workgroup that cannot start (a
                    // co-tenant holds its place) and polling flat out -- 10^5 lanes of uncached loads --
                    // starves the very kernel that has to finish first (seen: seconds instead of
                    // microseconds beside k_score)
                    if (spins < 32) __builtin_amdgcn_s_sleep(1);
                    else if (spins < 256) __builtin_amdgcn_s_sleep(16);
                    else __builtin_amdgcn_s_sleep(127);
                }
            }
            __hip_atomic_store(mine, (excl + tot) | LB_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        base[tid] = (u64)gbelow + excl;
    }
    if (PREFETCH == 2) load_tile(tile1, nkey);
    if (TICKETS && tid == THREADS - 1) ticket_s[0] = ahead;
    __syncthreads();
    // digit-contiguous runs out to global
    for (u32 j = tid; j < tile_n; j += THREADS) {
        const u64 k = stage[j];
        const u32 d = digit_of(k, pass);
        // (counts that do not match the keys -- a caller overwriting them mid-sort -- must end as a
        // poisoned result, not as a write outside the buffer)
        const u64 at = base[d] + (j - tile_off[d]);
        if (at < n) dst[at] = k; else atomicOr(err, 2u);
    }
    if (TICKETS) {
        tile = tile1 = ticket_s[0];
    } else {
        tile = tile1;
        tile1 += gridDim.x;
    }
    }  // next tile of this workgroup
}

// ---- positions: prefix sum of 2^(key - log2 n_global) ------------------------------------------
// Exclusive prefix of `v` over the workgroup (and the workgroup total).  The exclusive value is
// shuffled, never formed as inclusive - v: the weights span 2^-20..2^90, and that subtraction
// cancels catastrophically when one lane's value dwarfs everything before it.
__device__ __forceinline__ double block_exclusive_scan(double v, double *wave_tot, double &block_total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    double excl = __shfl_up(inc, 1);
    if (lane == 0) excl = 0.0;
    __syncthreads();  // protect wave_tot from the previous use
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    double off = 0.0, tot = 0.0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        const double t = wave_tot[w];
        if (w < wid) off += t;
        tot += t;
    }
    block_total = tot;
    return off + excl;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_weight_sums(const double *__restrict__ keys, u64 n, u64 chunk,
                                                               double logn, double *__restrict__ partials) {
    __shared__ double wave_tot[SCAN_THREADS / 64];
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    double s = 0.0;
    for (u64 i = lo + threadIdx.x; i < hi; i += SCAN_THREADS) s += exp2(keys[i] - logn);
    double tot;
    (void)block_exclusive_scan(s, wave_tot, tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_partials(double *partials, u32 nb) {
    // nb <= SCAN_MAX_BLOCKS: a single workgroup, sequential per lane then a block scan
    __shared__ double wave_tot[SCAN_THREADS / 64];
    const u32 per = (nb + SCAN_THREADS - 1) / SCAN_THREADS;
    const u32 lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    double s = 0.0;
    for (u32 i = lo; i < hi; ++i) s += partials[i];
    double tot;
    double run = block_exclusive_scan(s, wave_tot, tot);
    for (u32 i = lo; i < hi; ++i) {
        const double v = partials[i];
        partials[i] = run;  // exclusive
        run += v;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) k_positions(const double *__restrict__ keys, u64 n, u64 chunk,
                                                             double logn, const double *__restrict__ partials,
                                                             const u32 *__restrict__ err, double *__restrict__ pos) {
    __shared__ double wave_tot[SCAN_THREADS / 64];
    // A lane scans SCAN_ITEMS consecutive weights, but global memory is read and written with
    // consecutive lanes on consecutive elements (a lane-contiguous 64-byte access pattern makes every
    // load touch 64 cache lines: that alone held this kernel at 2.5 TB/s); the tile changes shape in
    // LDS, one pad word per 8 so that the lane-contiguous side is conflict-free.
    static_assert(SCAN_ITEMS == 8, "pad of the LDS tile");
    __shared__ double tile[SCAN_TILE + SCAN_TILE / 8];
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    // a sort pass that gave up waiting (k_onesweep) must not pass for a result
    double carry = *err ? __longlong_as_double(0x7FF8000000000000ll) : partials[blockIdx.x];
    for (u64 t0 = lo; t0 < hi; t0 += SCAN_TILE) {
        double k[SCAN_ITEMS];
#pragma unroll
        for (int r = 0; r < SCAN_ITEMS; ++r) {
            const u64 i = t0 + (u64)(r * SCAN_THREADS) + threadIdx.x;
            k[r] = i < hi ? keys[i] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < SCAN_ITEMS; ++r) {
            const u32 e = r * SCAN_THREADS + threadIdx.x;
            const u64 i = t0 + e;
            tile[e + (e >> 3)] = i < hi ? exp2(k[r] - logn) : 0.0;
        }
        __syncthreads();
        double *mine = tile + threadIdx.x * (SCAN_ITEMS + 1);  // == pad(threadIdx.x * 8 + j) for SCAN_ITEMS == 8
        double w[SCAN_ITEMS];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) {
            s += mine[j];
            w[j] = s;  // inclusive within the lane
        }
        double tot;
        const double excl = block_exclusive_scan(s, wave_tot, tot);
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; ++j) mine[j] = carry + (excl + w[j]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SCAN_ITEMS; ++r) {
            const u32 e = r * SCAN_THREADS + threadIdx.x;
            const u64 i = t0 + e;
            if (i < hi) pos[i] = tile[e + (e >> 3)];
        }
        carry += tot;
        __syncthreads();  // the next tile overwrites `tile`
    }
}

// ---- ml2p2rank for a batch ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_query(const double *__restrict__ keys, const double *__restrict__ pos, u64 n,
                                               const double *__restrict__ x, u64 t, double *__restrict__ out_pos,
                                               u64 *__restrict__ out_cnt) {
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < t; q += (u64)gridDim.x * blockDim.x) {
        const double v = x[q];
        u64 lo = 0, hi = n;  // first index with keys[i] > v  (bisect_right)
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (keys[mid] > v) hi = mid; else lo = mid + 1;
        }
        if (out_cnt) out_cnt[q] = lo;
        out_pos[q] = lo ? pos[lo - 1] : 0.0;
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The onesweep variant in use: workgroup shape (threads x keys per lane = tile) and where the next
// tile's keys are requested (0: at the top of its own iteration, 1: before the look-back, 2: after
// it).  MC_SORT_VARIANT (1: 256 x 24) / MC_SORT_TICKETS are read once; results never depend on them.  Only shapes
// that (a) compile without register spills and (b) fill the SIMDs evenly are offered: the fixed tile
// assignment needs the occupancy the API reports to be the occupancy the hardware gives, and a
// workgroup of 6 waves at 3 waves per SIMD (384 x 16) is reported as two per CU but placed as one --
// half the grid then never starts and the other half spins to the bound.
typedef void (*onesweep_fn)(const Plan *, int, const u64 *, u64 *, u64 *, u64, u32, const u32 *, u32 *, u64 *);
struct SortVariant {
    onesweep_fn fn;
    int threads, items, prefetch;
};
template <int T, int I, int W>
SortVariant onesweep_pick() {
    // (the prefetching forms of the kernel -- PREFETCH 1 and 2 -- are kept in the source for the next
    // compiler: with this one they spill or gain nothing, and none is instantiated)
    return SortVariant{k_onesweep<T, I, 0, W>, T, I, 0};
}
const SortVariant &sort_variant() {
    static const SortVariant v = [] {
        // default: tiles by ticket -- forward progress under ANY residency (other streams, other processes on
        // the GPU).  MC_SORT_EXCLUSIVE=1 selects the fixed tile assignment, ~17 % faster, for a caller that
        // guarantees the GPU to itself while the sort runs (bench.py does, and says so in its output).
        const int exclusive = getenv("MC_SORT_EXCLUSIVE") ? atoi(getenv("MC_SORT_EXCLUSIVE")) : 0;
        const int which = !exclusive ? 8 : getenv("MC_SORT_VARIANT") ? atoi(getenv("MC_SORT_VARIANT")) : 0;
        // (spill-free shapes only: the ones that needed scratch memory were slower, and the one run of such a
        // kernel beside k_score on a second queue faulted.  512 x 12 at 4 waves per SIMD: 0.553 ms per
        // build of 1e7 keys against 0.609 ms for 256 x 24 at 2; both with prefetch lose or spill)
        switch (which) {
        case 1: return onesweep_pick<256, 24, 2>();    // 6144 keys, 4 waves x 252 VGPRs, two workgroups per CU
        case 8: return SortVariant{k_onesweep<512, 12, 0, 4, true>, 512, 12, 0};  // tiles by ticket (shared GPU)
        default: return onesweep_pick<512, 12, 4>();   // 6144 keys, 8 waves x 128 VGPRs, two workgroups per CU
        }
    }();
    return v;
}

// Workgroups of k_onesweep the current device holds at once (its look-back relies on the whole grid
// being resident): the occupancy API's answer, capped by what the kernel's LDS allows.
int onesweep_resident_workgroups(int *out) {
    static int cached[64] = {0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
        int per_cu = 0, n_cu = 0;
        hipFuncAttributes attr;
        const void *fn = (const void *)sort_variant().fn;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, sort_variant().threads, 0));
        HIP_TRY(hipFuncGetAttributes(&attr, fn));
        HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        const int by_lds = attr.sharedSizeBytes ? (int)(160 * 1024 / attr.sharedSizeBytes) : per_cu;
        if (by_lds < per_cu) per_cu = by_lds;
        if (getenv("MC_SORT_WG_PER_CU") && atoi(getenv("MC_SORT_WG_PER_CU")) > 0 &&
            atoi(getenv("MC_SORT_WG_PER_CU")) < per_cu)
            per_cu = atoi(getenv("MC_SORT_WG_PER_CU"));
        if (per_cu < 1 || n_cu < 1) {
            mc_set_error("k_onesweep does not fit a compute unit (occupancy %d, %d CUs)", per_cu, n_cu);
            return MC_ERR_HIP;
        }
        cached[dev] = per_cu * n_cu;
    }
    *out = cached[dev];
    return MC_OK;
}

u32 sort_tiles(u64 n, u64 tile) { return (u32)((n + tile - 1) / tile); }
u32 scan_blocks(u64 n) {
    u64 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (u32)(tiles < (u64)SCAN_MAX_BLOCKS ? (tiles ? tiles : 1) : SCAN_MAX_BLOCKS);
}

}  // namespace

extern "C" size_t mc_rank_workspace_bytes(uint64_t n) {
    return align256(n * sizeof(u64)) + align256(HIST_ALL_WORDS * sizeof(u32)) +
           align256((size_t)2 * sort_tiles(n, SORT_TILE_MIN) * 256 * sizeof(u64)) + align256(sizeof(Plan)) +
           align256(SCAN_MAX_BLOCKS * sizeof(double));
}

extern "C" int mc_rank_build(const double *keys_in, uint64_t n, uint64_t n_global, double *keys_sorted,
                             double *positions, void *workspace, size_t workspace_bytes, void *stream) {
    if (n == 0) return MC_OK;
    if (!keys_in || !keys_sorted || !positions || !workspace || n_global < n || keys_in == keys_sorted ||
        (const void *)positions == (const void *)keys_sorted) {
        mc_set_error("mc_rank_build: bad argument");
        return MC_ERR_ARG;
    }
    if (n >= 0xFFFFFFFFull) {
        mc_set_error("mc_rank_build: n must be below 2^32 per GPU (got %llu)", (unsigned long long)n);
        return MC_ERR_ARG;
    }
    if (workspace_bytes < mc_rank_workspace_bytes(n)) {
        mc_set_error("mc_rank_build: workspace of %zu bytes, need %zu", workspace_bytes, mc_rank_workspace_bytes(n));
        return MC_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)workspace;
    SortWs ws;
    ws.temp = (u64 *)p;
    p += align256(n * sizeof(u64));
    const SortVariant &sv = sort_variant();
    const u32 tiles = sort_tiles(n, (u64)sv.threads * sv.items);
    const size_t lb_bytes = (size_t)2 * tiles * 256 * sizeof(u64);
    ws.hist = (u32 *)p;
    p += align256(HIST_ALL_WORDS * sizeof(u32));
    ws.lookback = (u64 *)p;
    p += align256(lb_bytes);
    ws.plan = (Plan *)p;
    p += align256(sizeof(Plan));
    ws.partials = (double *)p;
    u32 *err = ws.hist + 8 * 256 + 8;

    const u64 *in = (const u64 *)keys_in;
    u64 *A = (u64 *)keys_sorted, *B = ws.temp;
    // histograms, error word and both look-back regions are contiguous: one clear
    HIP_TRY(hipMemsetAsync(ws.hist, 0, align256(HIST_ALL_WORDS * sizeof(u32)) + lb_bytes, s));
    {
        u64 want = (n + 255) / 256;
        u32 g = (u32)(want < 2048 ? want : 2048);
        u64 hwant = (n + 2047) / 2048;
        u32 gh = (u32)(hwant < 1024 ? (hwant ? hwant : 1) : 1024);
        hipLaunchKernelGGL(k_hist_all, dim3(gh), dim3(256), 0, s, in, (u64)n, ws.hist);
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, ws.hist, (u64)n, ws.plan);
        hipLaunchKernelGGL(k_copy_if_unsorted, dim3(g), dim3(256), 0, s, ws.plan, in, A, (u64)n);
    }
    int resident = 0;
    {
        int rc = onesweep_resident_workgroups(&resident);
        if (rc != MC_OK) return rc;
    }
    const u32 os_grid = tiles < (u32)resident ? tiles : (u32)resident;
    for (int pass = 0; pass < 8; ++pass)
        hipLaunchKernelGGL(sv.fn, dim3(os_grid), dim3(sv.threads), 0, s, (const Plan *)ws.plan, pass, in, A, B, (u64)n,
                           tiles, (const u32 *)ws.hist, err, ws.lookback);
    // positions
    const double logn = std::log2((double)n_global);
    const u32 sb = scan_blocks(n);
    const u64 stiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const u64 schunk = ((stiles + sb - 1) / sb) * SCAN_TILE;
    hipLaunchKernelGGL(k_weight_sums, dim3(sb), dim3(SCAN_THREADS), 0, s, (const double *)keys_sorted, (u64)n, schunk,
                       logn, ws.partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(SCAN_THREADS), 0, s, ws.partials, sb);
    hipLaunchKernelGGL(k_positions, dim3(sb), dim3(SCAN_THREADS), 0, s, (const double *)keys_sorted, (u64)n, schunk,
                       logn, (const double *)ws.partials, (const u32 *)err, positions);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

// ---- mc_rank_build for callers that cannot allocate --------------------------------------------
// The workspace of mc_rank_build lives in this file-scope device array instead of a caller's buffer.
// Host code never passes the array's NAME to a kernel or a memset (on the host it is a stand-in, not a
// device address): the address is asked for once per device with hipGetSymbolAddress, with that device
// current, and kept; every GPU the library runs on holds its own copy of the array at its own address
// until the process ends.
static __device__ __align__(256) unsigned char mc_rank_scratch[MC_RANK_NOALLOC_SCRATCH_BYTES];

namespace {
std::mutex g_scratch_mutex;
void *g_scratch_ptr[64] = {nullptr};  // device address of mc_rank_scratch, per device

int rank_scratch_for_current_device(void **out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        mc_set_error("mc_rank_build_noalloc: device %d is outside the supported range 0..63", dev);
        return MC_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(mc_rank_scratch)));
        if (!p) {
            mc_set_error("mc_rank_build_noalloc: hipGetSymbolAddress returned a null address on device %d", dev);
            return MC_ERR_HIP;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return MC_OK;
}
}  // namespace

extern "C" uint64_t mc_rank_noalloc_max_keys(void) { return MC_RANK_NOALLOC_MAX_KEYS; }

extern "C" int mc_rank_build_noalloc(const double *keys_in, uint64_t n, uint64_t n_global, double *keys_sorted,
                                     double *positions, void *stream) {
    if (n > MC_RANK_NOALLOC_MAX_KEYS) {
        mc_set_error("mc_rank_build_noalloc: %llu keys, the built-in scratch holds at most %llu (use mc_rank_build "
                     "with a caller workspace beyond that)",
                     (unsigned long long)n, (unsigned long long)MC_RANK_NOALLOC_MAX_KEYS);
        return MC_ERR_ARG;
    }
    if (n && mc_rank_workspace_bytes(n) > (size_t)MC_RANK_NOALLOC_SCRATCH_BYTES) {
        mc_set_error("mc_rank_build_noalloc: %llu keys need %zu bytes of scratch, the array holds %llu",
                     (unsigned long long)n, mc_rank_workspace_bytes(n),
                     (unsigned long long)MC_RANK_NOALLOC_SCRATCH_BYTES);
        return MC_ERR_WORKSPACE;
    }
    void *scratch = nullptr;
    {
        int rc = rank_scratch_for_current_device(&scratch);
        if (rc != MC_OK) return rc;
    }
    // every call on this GPU shares the array: mc_rank_build clears the words that must start at zero
    // (histograms, error word, look-back words) on `stream` before its first kernel
    return mc_rank_build(keys_in, n, n_global, keys_sorted, positions, scratch, MC_RANK_NOALLOC_SCRATCH_BYTES, stream);
}

extern "C" int mc_rank_query(const double *keys_sorted, const double *positions, uint64_t n, const double *x,
                             uint64_t t, double *out_pos, uint64_t *out_cnt, void *stream) {
    if (t == 0) return MC_OK;
    if (!x || !out_pos || (n && (!keys_sorted || !positions))) {
        mc_set_error("mc_rank_query: bad argument");
        return MC_ERR_ARG;
    }
    u64 want = (t + 255) / 256;
    u32 g = (u32)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_query, dim3(g), dim3(256), 0, (hipStream_t)stream, keys_sorted, positions, (u64)n, x, (u64)t,
                       out_pos, (u64 *)out_cnt);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}
