// The key sort behind mc_rank_build (gfx950): replaces `minus_log_prob_list.sort()` of
// MonteCarloLib.__init__, reference lib4mc/MonteCarloLib.py:11.
//
// Sampled -log2 p are not generic keys.  A password model puts most of its mass on few strings, so most
// samples REPEAT a value (headline model: 1e7 samples hold 1.7e5 distinct values, 76 % of the samples are
// copies of 240 of them), and the sort is keys-only: equal keys are indistinguishable, nothing has to be
// stable.  The sort is built for that and degrades gracefully to any key set:
//
//   k_sniff    one workgroup counts a sample of 16384 keys and lists the values seen at least 5 times ("heavy",
//              <= 1024, sorted).  From 2^26 keys on the sample is 2^19 keys, counted by 32 workgroups into a
//              table in the workspace (k_sniff_count / _hist / _pick / _sort), the list holds up to 3072 values
//              and comes with a hash set in which a value lies in its home slot or the next one.
//   k_strip    one read of all keys: a key equal to a heavy value is COUNTED, not moved (lookup in an LDS hash
//              set); the others (the "tail") are written compacted, region by region, and their top-byte
//              histograms are taken on the way.  No heavy values found: nothing is copied.
//   levels     MSD radix passes over the tail, 8 bits each, most significant VARYING bits first: k_count
//              takes per-segment digit histograms and the OR / AND of every bin's keys, k_scatter moves the keys
//              into their bins and plans the next level: a bin of <= 4096 keys (or several neighbours that fit
//              together) becomes a "finish tile", a larger bin whose keys are all equal is cut into finish
//              tiles as it is, any other larger bin becomes a segment of the next level, its digit placed at
//              its own highest varying bit.  Destinations inside a bin are reserved with one atomic add per
//              (tile, bin) -- the order inside a bin is irrelevant -- so no workgroup ever waits for another
//              one: there is no look-back chain, no spinning, no residency assumption; every kernel makes
//              progress beside any other work on the GPU.
//   k_deep     (up to 2^25 keys) two launched levels; what is still too large then is finished by one workgroup
//              per segment through all its remaining levels.
//   k_finish   sorts a tile of <= 4096 keys in LDS: buckets them linearly over [min, max] (4096 buckets, ~1
//              key each), ranks inside small buckets by counting, larger ones by one wave, and a bucket beyond
//              that over its own extremes in a later round; writes every key at its final index = its rank in
//              the tail + the copies of smaller heavy values.
//   k_fill     writes the runs of the heavy values into the gaps left for them.
//
// All hand-offs between workgroups go through kernel boundaries on the stream.  Every store that depends on a
// count is bounds-checked and every table has a capacity check; a violation sets the error word (the caller
// poisons its result) instead of writing outside a buffer.
#include "mc_sort.h"

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

constexpr int CAP = 4096;          // keys a finish tile sorts in LDS; larger bins go one level deeper
constexpr int TILE = 4096;         // keys per count / scatter tile
constexpr int WG = 512;            // threads of the strip / count / scatter / finish workgroups
constexpr int ITEMS = TILE / WG;   // keys per lane and tile
constexpr int NWAVE = WG / 64;
constexpr int STRIP_SLOTS = 512;   // k_strip workgroups a GPU holds at once (104 VGPRs: two a CU): regions are sized for ONE round of them
constexpr int SNIFF_S = 16384, SNIFF_SLOTS = 16384, SNIFF_MIN = 5, SNIFF_THREADS = 1024, SNIFF_PROBES = 8;
constexpr int H_MAX = 1024, HSET = 2048;
// Large inputs (n >= BIG_N) look for more heavy values in a larger sample, counted over several workgroups in a
// table in the workspace: at 1.25e9 keys the 16384-key sample finds ~450 values (60-80 % of the keys) and the tail
// then pays 12 B a key and level; 3072 values cover 82-96 %.
constexpr int H_BIG = 3072, HSET_BIG = 6144;       // (6144 slots of 12 B: two k_strip workgroups still share a CU)
constexpr int BIG_S = 1 << 19, BIG_WGS = BIG_S / SNIFF_S, BIG_MIN = 6;
constexpr u32 G_SLOTS = 1u << 20, G_PROBES = 64, CNT_BINS = 1024, NONE = 0xFFFFFFFFu;
constexpr u64 BIG_N = 1ull << 26, BIG_LAYOUT_N = 1ull << 22;
constexpr int COPIES = 4;          // lane-sliced copies of LDS histograms: lanes l, l+4, ... share one
constexpr int MAX_LEVELS = 8;      // 64 bits / 8
constexpr int BINS = 4096;         // buckets of a finish tile
constexpr int SMALL_GROUP = 16;    // bucket sizes ranked by counting, lane per key
constexpr u64 EMPTY = ~0ull;

enum { ERR_SEG = 1u, ERR_TILE = 2u, ERR_FT = 4u, ERR_RANGE = 8u, ERR_DEBUG = 16u, ERR_LEVEL = 32u };
enum { SRC_IN = 0, SRC_B = 1, SRC_C = 2 };

struct Ctl {
    u32 err;
    u32 n_heavy;
    u32 n_tail;
    u32 levels_used;
    u64 or_all, nor_all;  // OR of the tail keys, OR of their complements (AND = ~nor_all)
    u32 nseg[MAX_LEVELS + 1];
    u32 ntile[MAX_LEVELS + 1];
    u32 nft;
    u32 segs_total;
    u32 deep_top;  // segments handed to k_deep's workgroups by their own plans (entries of the deep arena)
    u32 n_picked;  // large inputs: sample values at or above the bar (before they are sorted)
    u32 bar;       // ... and the bar
    // the heavy values of one sort serve the next sorts of as many keys in the same workspace (k_sniff): when the
    // list was made (magic = a list is there, for how many keys), the tail it left then, how often it has served
    u32 list_magic, list_n, list_tail, list_age;
};
constexpr u32 LIST_MAGIC = 0x4D43484Cu, LIST_MAX_AGE = 63;

struct Seg {
    u32 off, size, shift, pad;
};
struct HistRow {
    u32 cnt[256];   // keys per bin
    u32 cur[256];   // scatter cursor of the bin (reservations)
    u64 orv[256];   // OR of the bin's keys
    u64 norv[256];  // OR of their complements; both 0 = not collected (root level)
};
struct TileRef {
    u32 seg, k;
};
struct FTile {
    u32 off, size, src, pad;
};

struct Ws {
    Ctl *ctl;
    u64 *heavy;      // [H_MAX] sorted heavy values
    u32 *hcnt;       // [H_MAX] their copies
    u32 *gap;        // [H_MAX + 1] tail keys with exactly g heavy values below them
    HistRow *root;   // [2]: digit at bits 48..55, digit at bits 56..63 (tail keys)
    u32 *rcnt;       // [regions] tail keys of each region
    Seg *segs[2];
    HistRow *rows[2];
    TileRef *tiles[2];
    FTile *ft;
    const u64 *in;
    u64 *A, *B, *C;
    u32 n, region, regions;
    u32 cap_seg, cap_tile, cap_ft;
    u32 raw_f64;     // the input is raw fp64 bit patterns of ANY sign: every read of `in` maps them to words whose
                     // unsigned order is the numeric order (key_of), so the caller needs no pass of its own for that
    u32 reuse_list;  // k_sniff may take the heavy values the workspace holds from its last sort of n keys
    u32 *clear_from; // k_sniff clears the control tables itself (no memset launch before it): from here ...
    u32 clear_words; // ... this many words; 0: the host has done it
    // large inputs only (null otherwise)
    u64 *gk;         // [G_SLOTS] sample values (EMPTY = free)
    u32 *gc;         // [G_SLOTS] their counts in the sample
    u32 *ghist;      // [CNT_BINS] values per count (counts capped at CNT_BINS - 1)
    u64 *glist;      // [H_BIG] the values at or above the bar, unsorted
    u32 *glcnt;      // [H_BIG] their counts in the sample
    u64 *hkg;        // [HSET_BIG + 1] k_strip's hash set of the heavy values, built once
    u32 *hig;        // [HSET_BIG + 1] slot -> index of the heavy value (NONE = free)
    u32 *hpg;        // [H_BIG + 1] copies of the heavy values below index i
};

__device__ __forceinline__ u32 hset_hash(u64 k) { return (u32)((k * 0x9E3779B97F4A7C15ull) >> 40); }

// fp64 bits -> a word whose unsigned order is the numeric order (mc_sort_order_word, mc_sort.h)
__device__ __forceinline__ u64 key_of(u64 b) { return mc_sort_order_word(b); }

// exclusive prefix of v over the workgroup (every thread calls); `wtot` holds NWAVE words
__device__ __forceinline__ u32 wg_excl_scan(u32 v, u32 *wtot, u32 &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // wtot may still be read from the previous call
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

__device__ __forceinline__ u64 wave_or(u64 v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v |= __shfl_xor(v, o);
    return v;
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
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- heavy values -------------------------------------------------------------------------------------
// One workgroup: SNIFF_S keys (blocks of 16 consecutive keys, evenly spaced) are counted in an LDS hash table of
// 32-bit tags (a 64-bit table for that many samples would not fit); the slots seen SNIFF_MIN times or more (at
// most H_MAX: the bar is raised until they fit) are the heavy candidates; a second look at the same keys lists
// the key behind each such slot (the first one that claims it: two values sharing a tag cost one of them its
// place, nothing else), and the list is written in ascending order.
__device__ __forceinline__ u32 sniff_mix(u64 k) {
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    u32 x = lo ^ (hi * 0x9E3779B1u);
    x ^= x >> 15;
    return x * 0x85EBCA6Bu;
}

constexpr int SNIFF_BLOCK = 16;  // consecutive keys per sampled block

// One workgroup's SNIFF_S sample keys (block `block0 + i` of SNIFF_BLOCK keys starts at (block0 + i) * bstride),
// counted per slot of the LDS tag table; slot[q] = the slot of key q, or 0xFFFFFFFF when it found no place.
template <int PER>
__device__ __forceinline__ void sniff_sample(const u64 *in, bool raw, u64 bstride, u32 block0, u32 *tag, u32 *tc,
                                             u64 (&k)[PER], u32 (&slot)[PER]) {
    // (blocks of consecutive keys: 1024 pages touched; every key at its own stride was one address translation
    // per key for this one compute unit, most of the kernel's time)
    constexpr int BLOCK = SNIFF_BLOCK;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 block = (u32)((q * SNIFF_THREADS + tid) / BLOCK);
        k[q] = in[(u64)(block0 + block) * bstride + (u32)(tid % BLOCK)];
    }
    if (raw) {  // (uniform)
#pragma unroll
        for (int q = 0; q < PER; ++q) k[q] = key_of(k[q]);
    }
    // every key tries its home slot first, all PER attempts in flight together (one after the other they were PER
    // dependent LDS round trips); only a key that finds another value there walks on
    u32 tg[PER], old[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 x = sniff_mix(k[q]);
        tg[q] = x | 1u;
        slot[q] = x >> 18;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) old[q] = atomicCAS(&tag[slot[q]], 0u, tg[q]);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        // a sample that finds no place within SNIFF_PROBES slots is dropped: with few repeated values the table
        // fills up (as many samples as slots) and an unbounded probe would walk most of it for every sample
        u32 s = slot[q];
        bool placed = old[q] == 0u || old[q] == tg[q];
        for (int probe = 1; !placed && probe < SNIFF_PROBES; ++probe) {
            s = (s + 1) & (SNIFF_SLOTS - 1);
            const u32 o = atomicCAS(&tag[s], 0u, tg[q]);
            placed = o == 0u || o == tg[q];
        }
        slot[q] = placed ? s : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (slot[q] != 0xFFFFFFFFu) atomicAdd(&tc[slot[q] >> 1], 1u << (16 * (slot[q] & 1u)));
}

__global__ void __launch_bounds__(SNIFF_THREADS) k_sniff(Ws w) {
    constexpr int PER = SNIFF_S / SNIFF_THREADS;
    static_assert(SNIFF_SLOTS == 16384, "slot = top 14 bits of the mix");
    __shared__ u32 tag[SNIFF_SLOTS];          // 0 = empty
    __shared__ u32 tc[SNIFF_SLOTS / 2];       // two 16-bit counts per word
    __shared__ u32 claim[SNIFF_SLOTS / 32];
    __shared__ u64 list[H_MAX];
    __shared__ u32 nlist;
    const u32 n = w.n;
    const int tid = threadIdx.x;
    // The heavy values of the LAST sort of n keys in this workspace are taken again when there are any: keys from
    // the same source (sampled scores of one model, the scores of one test set) repeat the same values, and ANY
    // ascending list of distinct values gives a correct sort -- a stale list only leaves a longer tail.  So the
    // search is run when the workspace holds no list for n keys, when the last sort failed, when the tail has grown
    // by a quarter over the one the list left when it was made (another source), when the list covers less than a
    // quarter of the keys (little to lose, and a source with more repeats may have taken over), and every 64th time.
    // 21 -> 5 us.
    __shared__ u32 keep_h, keep_tail, keep_age;
    if (tid == 0) {
        const Ctl c = *w.ctl;
        const u32 base = c.list_age == 0 ? c.n_tail : c.list_tail;  // (the list's own sort has told its tail by now)
        const bool ok = w.reuse_list && c.list_magic == LIST_MAGIC && c.list_n == n && c.err == 0 && c.n_heavy > 0 &&
                        c.n_heavy <= (u32)H_MAX && c.list_age < LIST_MAX_AGE && c.n_tail <= n &&
                        (u64)c.n_tail * 4 <= (u64)base * 5 + n / 32 && (u64)c.n_tail * 4 <= (u64)n * 3;
        keep_h = ok ? c.n_heavy : 0u;
        keep_tail = base;
        keep_age = c.list_age + 1;
    }
    __syncthreads();
    u32 keep = keep_h;
    {
        int unordered = 0;
        for (u32 i = tid + 1; i < keep; i += SNIFF_THREADS) unordered |= !(w.heavy[i - 1] < w.heavy[i]);
        if (__syncthreads_or(unordered)) keep = 0;  // (never a list of ours)
    }
    // the control tables (counters, histograms, the heavy lists) start at zero: this workgroup runs alone and first
    {
        const u32 *keep0 = (const u32 *)w.heavy, *keep1 = keep0 + 2 * keep;
        for (u32 i = tid; i < w.clear_words; i += SNIFF_THREADS) {
            u32 *p = w.clear_from + i;
            if (p < keep0 || p >= keep1) *p = 0;
        }
    }
    if (n < (u32)SNIFF_S * 4) return;  // small inputs: no heavy values (n_heavy stays 0)
    if (keep) {
        __syncthreads();
        if (tid == 0) {
            w.ctl->n_heavy = keep;
            w.ctl->list_magic = LIST_MAGIC;
            w.ctl->list_n = n;
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
    sniff_sample<PER>(w.in, w.raw_f64 != 0, n / (SNIFF_S / SNIFF_BLOCK), 0u, tag, tc, k, slot);
    u32 bar = SNIFF_MIN;
    for (;;) {  // (uniform: every thread sees the same nlist)
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
    __syncthreads();
    if (tid == 0) nlist = 0;
    __syncthreads();
    bool over[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {  // (the reads of all PER first, then the claims)
        const u32 s = slot[q];
        over[q] = s != 0xFFFFFFFFu && ((tc[s >> 1] >> (16 * (s & 1u))) & 0xFFFFu) >= bar;
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
    __syncthreads();
    const u32 H = nlist;
    // the values are distinct: rank = position.  All lanes take part: a value's comparisons are split over
    // SNIFF_THREADS / H_MAX ... lanes when there are few values (one lane a value walked the whole list alone)
    {
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
        w.ctl->n_heavy = H;
        w.ctl->list_magic = LIST_MAGIC;  // (list_age = 0, list_tail = 0: the next call reads this sort's n_tail)
        w.ctl->list_n = n;
    }
}

// ---- strip: count the copies of heavy values, compact the tail ------------------------------------------
// Per tile of 8 keys a lane: all hash-table reads are issued together, the heavy keys are counted per table
// SLOT (translated to the heavy value's index once per workgroup), the tail keys of the 8 rounds take their
// places from ONE add per wave.  (The first version did each key's lookup, count and compaction one after the
// other, ~10 dependent LDS round trips per key: 88 us; instruction count and LDS latency, not memory, set the pace.)
__device__ __forceinline__ u32 strip_hash(u64 k) {
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    u32 x = lo ^ (hi * 0x9E3779B1u);
    x ^= x >> 15;
    return (x * 0x85EBCA6Bu) >> 21;  // 11 bits = HSET slots
}
__device__ __forceinline__ u32 strip_hash_big(u64 k) {  // HSET_BIG slots (not a power of two)
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    u32 x = lo ^ (hi * 0x9E3779B1u);
    x ^= x >> 15;
    return __umulhi(x * 0x85EBCA6Bu, (u32)HSET_BIG);
}

// One tile of k_strip.  FULL: every lane holds ITEMS keys (no validity tests at all: they were a branch per load
// and per key).  Heavy test without branches: the key's home slot and the next one are read together; only a key
// that meets two other values there walks on (rare: the table is at most half full, 14 % on the headline model).
template <bool FULL, bool BIG>
__device__ __forceinline__ void strip_tile(const u64 (&k)[ITEMS], u32 c0, u32 rn, u64 lo, u32 H, const u64 *hk,
                                           u32 *hs, u32 *h48, u32 *h56, u32 *roff, u64 *B, u64 &o, u64 &no) {
    constexpr u32 SLOTS = BIG ? (u32)HSET_BIG : (u32)HSET;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 cp = (u32)(lane & (COPIES - 1)) * 256u;
    const u64 lt = (1ull << lane) - 1ull;
    u64 tmask[ITEMS];
    u32 total = 0;
    if (H) {
        u32 s[ITEMS];
        u64 a[ITEMS], b[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            s[q] = BIG ? strip_hash_big(k[q]) : strip_hash(k[q]);
            a[q] = hk[s[q]];
            b[q] = hk[s[q] + 1];  // (slot HSET is a spare that stays empty; BIG: the last slot's neighbour)
        }
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const bool valid = FULL || c0 + q * WG + tid < rn;
            const bool real = valid && k[q] != EMPTY;  // (a key equal to the empty marker is never heavy)
            const bool hit_a = real && a[q] == k[q];
            // (BIG: a value lies in its home slot or the next one, or it is not in the set -- k_sniff_sort)
            const bool hit_b = real && !hit_a && (BIG || a[q] != EMPTY) && b[q] == k[q];
            // (a miss is decided when a slot is empty; the table wraps, so the spare slot decides nothing)
            bool walk = !BIG && real && !hit_a && !hit_b && a[q] != EMPTY && (b[q] != EMPTY || s[q] + 1 == SLOTS);
            bool heavy = hit_a || hit_b;
            u32 slot = s[q] + (hit_b ? 1u : 0u);
            if (!BIG && __ballot(walk)) {  // (rare)
                if (walk) {
                    u32 t = s[q] + 1 == SLOTS ? 0u : (s[q] + 2 == SLOTS ? 0u : s[q] + 2);
                    for (;;) {
                        const u64 v = hk[t];
                        if (v == k[q]) {
                            heavy = true;
                            slot = t;
                            break;
                        }
                        if (v == EMPTY) break;
                        t = t + 1 == SLOTS ? 0u : t + 1;
                    }
                }
            }
            if (heavy) atomicAdd(&hs[slot], 1u);
            tmask[q] = __ballot(valid && !heavy);
            total += (u32)__popcll(tmask[q]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            tmask[q] = __ballot(FULL || c0 + q * WG + tid < rn);
            total += (u32)__popcll(tmask[q]);
        }
    }
    if (total == 0) return;  // (wave-uniform)
    u32 base = 0;
    if (H) {  // compact (no heavy values: the region is its own tail and stays where it is)
        if (lane == 0) base = atomicAdd(roff, total);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    // top-byte histograms of the tail
    u64 kf = 0;  // the wave's first tail key
    bool got = false;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const bool tail = (tmask[q] >> lane) & 1ull;
        if (tail) {
            if (H) B[lo + base + (u32)__popcll(tmask[q] & lt)] = k[q];
            atomicAdd(&h48[cp + ((u32)(k[q] >> 48) & 255u)], 1u);
            o |= k[q];
            no |= ~k[q];
        }
        base += (u32)__popcll(tmask[q]);
        if (!got && tmask[q]) {
            kf = __shfl(k[q], __ffsll((long long)tmask[q]) - 1);
            got = true;
        }
    }
    // bits 56..63 are one value almost always: one add per wave then
    const u32 d56_0 = (u32)(kf >> 56);
    u64 differ = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) differ |= __ballot(((tmask[q] >> lane) & 1ull) && (u32)(k[q] >> 56) != d56_0);
    if (!differ) {
        if (lane == 0) atomicAdd(&h56[d56_0], total);
    } else {
#pragma unroll
        for (int q = 0; q < ITEMS; ++q)
            if ((tmask[q] >> lane) & 1ull) atomicAdd(&h56[(u32)(k[q] >> 56)], 1u);
    }
}

// BIG: the hash set was built once (k_sniff_sort) and is copied in; slot -> value index stays in memory.
template <bool BIG>
__global__ void __launch_bounds__(WG) k_strip(Ws w) {
    static_assert(HSET == 2048, "strip_hash yields 11 bits");
    constexpr u32 SLOTS = BIG ? (u32)HSET_BIG : (u32)HSET;
    __shared__ u64 hk[SLOTS + 1];         // (+1: the pair read of the last slot)
    __shared__ u32 hs[SLOTS + 1];         // copies counted per slot
    __shared__ u32 hi[BIG ? 1 : HSET];    // slot -> index of the heavy value
    __shared__ u32 h48[256 * COPIES], h56[256];
    __shared__ u32 roff;
    __shared__ u64 red[2 * NWAVE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 H = w.ctl->n_heavy;
    const u32 n = w.n;
    for (u32 i = tid; i <= SLOTS; i += WG) hk[i] = BIG ? w.hkg[i] : EMPTY;
    for (u32 i = tid; i <= SLOTS; i += WG) hs[i] = 0;
    for (int i = tid; i < 256 * COPIES; i += WG) h48[i] = 0;
    if (tid < 256) h56[tid] = 0;
    __syncthreads();
    if (!BIG) {
        for (u32 i = tid; i < H; i += WG) {
            const u64 v = w.heavy[i];
            u32 s = strip_hash(v);
            for (;;) {  // H <= HSET / 2 distinct values
                const u64 old = atomicCAS(&hk[s], EMPTY, v);
                if (old == EMPTY) {
                    hi[s] = i;
                    break;
                }
                s = (s + 1) & (HSET - 1);
            }
        }
        __syncthreads();
    }
    u64 o = 0, no = 0;
    u32 tail_total = 0;  // (thread 0)
    for (u32 r = blockIdx.x; r < w.regions; r += gridDim.x) {
        const u64 lo = (u64)r * w.region;
        const u32 rn = (u32)(n - lo < (u64)w.region ? n - lo : (u64)w.region);
        if (tid == 0) roff = 0;
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
            if (w.raw_f64) {  // (uniform)
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) k[q] = key_of(k[q]);
            }
            if (c0 + TILE < rn) {  // the next tile travels while this one is looked up
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = c0 + TILE + q * WG + tid;
                    nk[q] = w.in[lo + (j < rn ? j : rn - 1)];
                }
            }
            if (c0 + TILE <= rn) strip_tile<true, BIG>(k, c0, rn, lo, H, hk, hs, h48, h56, &roff, w.B, o, no);
            else strip_tile<false, BIG>(k, c0, rn, lo, H, hk, hs, h48, h56, &roff, w.B, o, no);
        }
        __syncthreads();
        if (tid == 0) {
            const u32 m = H ? roff : rn;
            w.rcnt[r] = m;
            tail_total += m;
        }
        __syncthreads();
    }
    o = wave_or(o);
    no = wave_or(no);
    if (lane == 0) {
        red[wid] = o;
        red[NWAVE + wid] = no;
    }
    __syncthreads();
    if (tid == 0) {
        u64 a = 0, b = 0;
        for (int i = 0; i < NWAVE; ++i) {
            a |= red[i];
            b |= red[NWAVE + i];
        }
        if (a | b) {
            atomicOr(&w.ctl->or_all, a);
            atomicOr(&w.ctl->nor_all, b);
        }
        if (tail_total) atomicAdd(&w.ctl->n_tail, tail_total);
    }
    if (tid < 256) {
        u32 v = 0;
#pragma unroll
        for (int c = 0; c < COPIES; ++c) v += h48[c * 256 + tid];
        if (v) atomicAdd(&w.root[0].cnt[tid], v);
        if (h56[tid]) atomicAdd(&w.root[1].cnt[tid], h56[tid]);
    }
    // per-slot counts -> per-value counts in LDS, then ONE coalesced add per value: adds scattered over the array
    // (slot order) are one memory-side request each and queue on its few lines (measured: 120 us for 266 values
    // x 611 workgroups)
    __syncthreads();
    static_assert(256 * COPIES >= H_MAX, "h48 is reused for the per-value counts");
    static_assert((HSET_BIG + 1) * 2 >= H_BIG, "the hash set's own memory is reused for the per-value counts");
    u32 *hc = BIG ? (u32 *)hk : h48;  // (free again: the keys are all looked up; h48 holds H_MAX words)
    if (BIG) {
        for (u32 i = tid; i < H; i += WG) hc[i] = 0;
        __syncthreads();
        for (u32 i = tid; i <= SLOTS; i += WG) {
            const u32 c = hs[i], idx = c ? w.hig[i] : NONE;
            if (idx < H) hc[idx] = c;
        }
    } else {
        for (u32 i = tid; i < SLOTS; i += WG)
            if (hk[i] != EMPTY) hc[hi[i]] = hs[i];
    }
    __syncthreads();
    for (u32 i = tid; i < H; i += WG)
        if (hc[i]) atomicAdd(&w.hcnt[i], hc[i]);
}

// ---- heavy values of a large input ------------------------------------------------------------------------
// BIG_S sample keys instead of 16384: BIG_WGS workgroups count SNIFF_S keys each as k_sniff does and add every
// slot's (value, count) to one table in the workspace; a histogram of the counts gives the bar that leaves at most
// H_BIG values; those are listed, sorted, and hashed ONCE into the set every k_strip workgroup copies.  Any list
// of distinct values is a correct list (a value's copies are counted exactly by k_strip, whatever the sample
// said): a tag shared by two values or a full probe window only costs a better choice.
__device__ __forceinline__ u32 g_hash(u64 k) { return (u32)((k * 0x9E3779B97F4A7C15ull) >> 44); }  // 20 bits
static_assert(G_SLOTS == 1u << 20, "g_hash yields 20 bits");

__global__ void __launch_bounds__(SNIFF_THREADS) k_sniff_count(Ws w) {
    constexpr int PER = SNIFF_S / SNIFF_THREADS;
    __shared__ u32 tag[SNIFF_SLOTS];
    __shared__ u32 tc[SNIFF_SLOTS / 2];
    __shared__ u32 claim[SNIFF_SLOTS / 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < SNIFF_SLOTS; i += SNIFF_THREADS) tag[i] = 0;
    for (int i = tid; i < SNIFF_SLOTS / 2; i += SNIFF_THREADS) tc[i] = 0;
    for (int i = tid; i < SNIFF_SLOTS / 32; i += SNIFF_THREADS) claim[i] = 0;
    __syncthreads();
    constexpr u32 BLOCKS = SNIFF_S / SNIFF_BLOCK;  // per workgroup
    u64 k[PER];
    u32 slot[PER];
    // (the host launches this for n >= BIG_LAYOUT_N only: a block's stride is far above its 16 keys)
    sniff_sample<PER>(w.in, w.raw_f64 != 0, w.n / (BLOCKS * gridDim.x), blockIdx.x * BLOCKS, tag, tc, k, slot);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 s = slot[q];
        if (s == 0xFFFFFFFFu || k[q] == EMPTY) continue;  // (a key equal to the empty marker is never heavy)
        const u32 bit = 1u << (s & 31u);
        if (atomicOr(&claim[s >> 5], bit) & bit) continue;  // one key speaks for its slot
        const u32 c = (tc[s >> 1] >> (16 * (s & 1u))) & 0xFFFFu;
        u32 g = g_hash(k[q]);
        for (u32 probe = 0; probe < G_PROBES; ++probe) {
            const u64 old = atomicCAS(&w.gk[g], EMPTY, k[q]);
            if (old == EMPTY || old == k[q]) {
                atomicAdd(&w.gc[g], c);
                break;
            }
            g = (g + 1) & (G_SLOTS - 1);
        }
    }
}

__global__ void __launch_bounds__(1024) k_sniff_hist(Ws w) {
    __shared__ u32 lh[CNT_BINS];
    static_assert(CNT_BINS == 1024, "one bin per thread");
    const u32 tid = threadIdx.x;
    lh[tid] = 0;
    __syncthreads();
    for (u32 i = blockIdx.x * 1024u + tid; i < G_SLOTS; i += gridDim.x * 1024u) {
        const u32 c = w.gc[i];
        if (c >= (u32)BIG_MIN) atomicAdd(&lh[c < CNT_BINS ? c : CNT_BINS - 1], 1u);
    }
    __syncthreads();
    if (lh[tid]) atomicAdd(&w.ghist[tid], lh[tid]);
}

__global__ void __launch_bounds__(1024) k_sniff_pick(Ws w) {
    __shared__ u32 wt[16];
    __shared__ u32 bar_s;
    const u32 tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    // values with a count of at least c, for c = CNT_BINS - 1 - tid: an inclusive scan from the top
    const u32 c = CNT_BINS - 1 - tid;
    u32 inc = w.ghist[c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o);
        if (lane >= (u32)o) inc += t;
    }
    if (lane == 63) wt[wid] = inc;
    if (tid == 0) bar_s = CNT_BINS;  // (no count qualifies: nothing is picked)
    __syncthreads();
    for (u32 i = 0; i < wid; ++i) inc += wt[i];
    // the lowest bar that leaves at most H_BIG values (counts are capped at CNT_BINS - 1, where fewer than
    // BIG_S / (CNT_BINS - 1) < H_BIG values can be: the bar never needs to go beyond the cap)
    if (c >= (u32)BIG_MIN && inc <= (u32)H_BIG) atomicMin(&bar_s, c);
    __syncthreads();
    const u32 bar = bar_s;
    for (u32 i = blockIdx.x * 1024u + tid; i < G_SLOTS; i += gridDim.x * 1024u) {
        const u32 v = w.gc[i];
        if ((v < CNT_BINS ? v : CNT_BINS - 1) >= bar) {
            const u32 p = atomicAdd(&w.ctl->n_picked, 1u);
            if (p < (u32)H_BIG) {
                w.glist[p] = w.gk[i];
                w.glcnt[p] = v;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) w.ctl->bar = bar;
}

// One workgroup: k_strip's hash set of the picked values, then the values in ascending order (bitonic network in
// LDS).  The set is built so that a lookup is two reads and never a walk: a value lies in its home slot or the
// next one; a value that finds both taken is left out of the list (it stays in the tail; at half load 9 % of
// the values, which go in by descending count so that the commonest are placed while the table is empty).
// (With open-ended probing every wave of k_strip walked for its unluckiest lane: 8.3 ms instead of 4.4 at 1.25e9.)
__global__ void __launch_bounds__(1024) k_sniff_sort(Ws w) {
    constexpr u32 M = 4096;
    static_assert(H_BIG <= M, "the network sorts M keys");
    __shared__ u64 a[M];
    __shared__ u64 hk[HSET_BIG + 1];
    __shared__ u32 hi[HSET_BIG + 1];
    __shared__ u32 kept;
    const u32 tid = threadIdx.x;
    const u32 picked = w.ctl->n_picked, bar = w.ctl->bar;
    const u32 P = picked < (u32)H_BIG ? picked : (u32)H_BIG;
    constexpr int PT = M / 1024;
    u32 cnt[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = q * 1024u + tid;
        a[i] = i < P ? w.glist[i] : EMPTY;  // (no picked value is EMPTY)
        cnt[q] = i < P ? w.glcnt[i] : 0u;
    }
    for (u32 i = tid; i <= (u32)HSET_BIG; i += 1024u) {
        hk[i] = EMPTY;
        hi[i] = NONE;
    }
    if (tid == 0) kept = 0;
    __syncthreads();
    for (int round = 0; round < 4; ++round) {  // counts of 64, 16, 4 bars and more, then the rest
        const u64 from = round < 3 ? (u64)bar << (6 - 2 * round) : 0ull;
        const u64 below = round ? (u64)bar << (8 - 2 * round) : ~0ull;
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const u32 i = q * 1024u + tid;
            if (i < P && cnt[q] >= from && cnt[q] < below) {
                const u64 v = a[i];
                const u32 s = strip_hash_big(v);
                if (atomicCAS(&hk[s], EMPTY, v) != EMPTY && atomicCAS(&hk[s + 1], EMPTY, v) != EMPTY) a[i] = EMPTY;
            }
        }
        __syncthreads();
    }
    for (u32 k2 = 2; k2 <= M; k2 <<= 1) {
        for (u32 j = k2 >> 1; j > 0; j >>= 1) {
            for (u32 t = tid; t < M / 2; t += 1024u) {
                const u32 lo = ((t / j) * 2u * j) + (t % j), hi2 = lo + j;
                const bool up = (lo & k2) == 0;
                const u64 x = a[lo], y = a[hi2];
                if ((x > y) == up) {
                    a[lo] = y;
                    a[hi2] = x;
                }
            }
            __syncthreads();
        }
    }
    for (u32 i = tid; i < M; i += 1024u) {
        const u64 v = a[i];
        if (v == EMPTY) continue;
        if (i + 1 == M || a[i + 1] == EMPTY) kept = i + 1;  // (one thread: the values are distinct)
        w.heavy[i] = v;
        const u32 s = strip_hash_big(v);
        hi[hk[s] == v ? s : s + 1] = i;
    }
    __syncthreads();
    for (u32 i = tid; i <= (u32)HSET_BIG; i += 1024u) {
        w.hkg[i] = hk[i];
        w.hig[i] = hi[i];
    }
    if (tid == 0) w.ctl->n_heavy = kept;
}

// hpg[i] = copies of the heavy values below index i (i <= H), for k_finish<BIG>
__global__ void __launch_bounds__(1024) k_heavy_prefix(Ws w) {
    constexpr int PT = H_BIG / 1024;
    static_assert(PT * 1024 == H_BIG, "whole heavy values per thread");
    __shared__ u32 wt[16];
    const u32 tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const u32 H = w.ctl->n_heavy;
    u32 c[PT], s = 0;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        c[q] = i < H ? w.hcnt[i] : 0u;
        s += c[q];
    }
    u32 inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o);
        if (lane >= (u32)o) inc += t;
    }
    if (lane == 63) wt[wid] = inc;
    __syncthreads();
    u32 ex = inc - s;
    for (u32 i = 0; i < wid; ++i) ex += wt[i];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        if (i <= H) w.hpg[i] = ex;
        ex += c[q];
    }
    if (tid == 1023 && H == (u32)H_BIG) w.hpg[H] = ex;
}

// ---- one MSD level --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_count(Ws w, int level) {
    __shared__ u32 lc[256];
    __shared__ u64 lor[256], lnor[256];
    const int tid = threadIdx.x;
    const int set = level & 1;
    if (w.ctl->err) return;  // a table overflowed in an earlier kernel: its entries are not to be trusted
    const u32 nt = w.ctl->ntile[level] < w.cap_tile ? w.ctl->ntile[level] : w.cap_tile;
    const u64 *src = (level & 1) ? w.C : w.B;
    for (u32 t = blockIdx.x; t < nt; t += gridDim.x) {
        const TileRef ref = w.tiles[set][t];
        const Seg sg = w.segs[set][ref.seg];
        const u32 k0 = ref.k * TILE;
        const u32 m = sg.size - k0 < (u32)TILE ? sg.size - k0 : (u32)TILE;
        if (tid < 256) {
            lc[tid] = 0;
            lor[tid] = 0;
            lnor[tid] = 0;
        }
        __syncthreads();
        const u64 *p = src + sg.off + k0;
        u64 k[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 j = q * WG + tid;
            k[q] = j < m ? p[j] : 0ull;
        }
        // (independent adds, all in flight together: a bin that many lanes share costs ~2 cycles per lane,
        // a dependent round trip per key costs far more)
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 j = q * WG + tid;
            if (j < m) {
                const u32 d = (u32)(k[q] >> sg.shift) & 255u;
                atomicAdd(&lc[d], 1u);
                atomicOr(&lor[d], k[q]);
                atomicOr(&lnor[d], ~k[q]);
            }
        }
        __syncthreads();
        if (tid < 256 && lc[tid]) {
            HistRow *row = &w.rows[set][ref.seg];
            atomicAdd(&row->cnt[tid], lc[tid]);
            atomicOr(&row->orv[tid], lor[tid]);
            atomicOr(&row->norv[tid], lnor[tid]);
        }
        __syncthreads();
    }
}

struct PlanLds {
    u32 cls[256], ccnt[256], cshift[256];
    u32 prev_ne[256];   // (inclusive max-scan) 1 + index of the last non-empty bin at or before b
    u32 next_bd[257];   // (suffix min-scan) first boundary bin at or after b; [256] = 256
    u32 flag[256];      // bin starts a packed finish tile
    u32 wtot[NWAVE];
    u32 ft_base, seg_base, tile_base, ok, n_big;
};

// Decide what becomes of every bin of one segment (run by ONE workgroup per segment; `coff` = the bins'
// offsets inside the segment, `seg_size` its keys): finish tiles, or segments and tiles of the next level.
// Bins of <= CAP / 2 keys are packed: neighbours whose offsets fall into the same window of CAP / 2 share a tile
// (it ends before two windows are over, so it holds < CAP keys); a bin of up to CAP keys is a tile of its own; a
// larger bin whose keys are all equal is cut into tiles as it lies; any other larger bin is a segment of the next
// level.  All of it in parallel over the 256 bins: a serial walk by one lane was the longest thing in the kernel.
// `deep` != nullptr (k_deep): the next level's segments go to that arena (level in `pad`), pl.seg_base / pl.n_big
// say where; no tiles are listed and no rows cleared -- the workgroup that planned them sorts them itself.
__device__ void plan_segment(const Ws &w, int level, u32 seg_off, u32 seg_size, u32 shift, const u32 *rcnt,
                             const u64 *rorv, const u64 *rnorv, const u32 *coff, u32 dst_id, PlanLds &pl,
                             Seg *deep = nullptr, u32 deep_cap = 0) {
    const int tid = threadIdx.x;
    const int next = (level + 1) & 1;
    u32 cls = 0, c = 0, csh = 0;  // 0 empty, 1 packed, 2 own tile, 3 large and all equal, 4 large
    if (tid < 256) {
        c = rcnt[tid];
        const u64 o = rorv[tid], no = rnorv[tid];
        const bool known = (o | no) != 0;
        const u64 diff = o ^ ~no;
        if (c) {
            if (c <= (u32)CAP / 2) cls = 1u;
            else if (c <= (u32)CAP) cls = 2u;
            else if ((known && diff == 0) || shift == 0) cls = 3u;
            else {
                cls = 4u;
                if (known) {
                    const int hb = 63 - __builtin_clzll(diff);
                    csh = hb > 7 ? (u32)(hb - 7) : 0u;
                } else {
                    csh = shift - 8;  // (root level only: shift is 56 or 48)
                }
            }
        }
        pl.cls[tid] = cls;
        pl.ccnt[tid] = c;
        pl.cshift[tid] = csh;
        pl.prev_ne[tid] = c ? (u32)tid + 1u : 0u;
    }
    if (tid == 0) pl.next_bd[256] = 256u;
    __syncthreads();
    // inclusive max-scan of prev_ne (Hillis-Steele over 256 entries)
    for (int o = 1; o < 256; o <<= 1) {
        u32 v = 0;
        if (tid < 256 && tid >= o) v = pl.prev_ne[tid - o];
        __syncthreads();
        if (tid < 256 && v > pl.prev_ne[tid]) pl.prev_ne[tid] = v;
        __syncthreads();
    }
    // a packed bin starts a tile when the non-empty bin before it is none, is not packed, or lies in another window
    if (tid < 256) {
        u32 f = 0;
        if (cls == 1u) {
            const u32 p = tid ? pl.prev_ne[tid - 1] : 0u;  // 1 + index of the previous non-empty bin
            f = 1u;
            if (p && pl.cls[p - 1] == 1u && coff[p - 1] / (CAP / 2) == coff[tid] / (CAP / 2)) f = 0u;
        }
        pl.flag[tid] = f;
        pl.next_bd[tid] = (f || cls >= 2u) ? (u32)tid : 256u;
    }
    __syncthreads();
    // suffix min-scan of next_bd
    for (int o = 1; o < 256; o <<= 1) {
        u32 v = 256u;
        if (tid < 256 && tid + o < 256) v = pl.next_bd[tid + o];
        __syncthreads();
        if (tid < 256 && v < pl.next_bd[tid]) pl.next_bd[tid] = v;
        __syncthreads();
    }
    // how many of each, and where they go
    const u32 is_ft = (tid < 256 && (pl.flag[tid] || cls == 2u)) ? 1u : 0u;
    const u32 n_chunk = cls == 3u ? (c + CAP - 1) / CAP : 0u;
    const u32 is_big = cls == 4u ? 1u : 0u;
    const u32 n_tl = cls == 4u ? (c + TILE - 1) / TILE : 0u;
    u32 tot_ft, tot_chunk, tot_big, tot_tl;
    const u32 ex_ft = wg_excl_scan(is_ft, pl.wtot, tot_ft);
    const u32 ex_chunk = wg_excl_scan(n_chunk, pl.wtot, tot_chunk);
    const u32 ex_big = wg_excl_scan(is_big, pl.wtot, tot_big);
    const u32 ex_tl = wg_excl_scan(n_tl, pl.wtot, tot_tl);
    if (tid == 0) {
        u32 ok = 1;
        pl.ft_base = atomicAdd(&w.ctl->nft, tot_ft + tot_chunk);
        if ((u64)pl.ft_base + tot_ft + tot_chunk > (u64)w.cap_ft) {
            atomicOr(&w.ctl->err, ERR_FT);
            ok = 0;
        }
        pl.seg_base = pl.tile_base = 0;
        pl.n_big = tot_big;
        if (tot_big) {
            if (level + 1 >= MAX_LEVELS) {  // cannot happen: the digit moves down by >= 8 bits per level
                atomicOr(&w.ctl->err, ERR_LEVEL);
                ok = 0;
            } else if (deep) {
                pl.seg_base = atomicAdd(&w.ctl->deep_top, tot_big);
                if ((u64)pl.seg_base + tot_big > (u64)deep_cap) {
                    atomicOr(&w.ctl->err, ERR_SEG);
                    ok = 0;
                }
                atomicMax(&w.ctl->levels_used, (u32)(level + 2));
                atomicAdd(&w.ctl->segs_total, tot_big);
            } else {
                pl.seg_base = atomicAdd(&w.ctl->nseg[level + 1], tot_big);
                pl.tile_base = atomicAdd(&w.ctl->ntile[level + 1], tot_tl);
                if ((u64)pl.seg_base + tot_big > (u64)w.cap_seg) {
                    atomicOr(&w.ctl->err, ERR_SEG);
                    ok = 0;
                }
                if ((u64)pl.tile_base + tot_tl > (u64)w.cap_tile) {
                    atomicOr(&w.ctl->err, ERR_TILE);
                    ok = 0;
                }
                atomicMax(&w.ctl->levels_used, (u32)(level + 2));
                atomicAdd(&w.ctl->segs_total, tot_big);
            }
        }
        pl.ok = ok;
    }
    __syncthreads();
    if (!pl.ok) return;  // (the error word is set: the result will be poisoned; write nothing out of range)
    if (is_ft) {
        // the tile runs from this bin to the next boundary bin; everything between is packed bins back to back
        const u32 nb = cls == 2u ? 256u : pl.next_bd[tid + 1];
        FTile f;
        f.off = seg_off + coff[tid];
        f.size = cls == 2u ? c : (nb < 256u ? coff[nb] : seg_size) - coff[tid];
        f.src = dst_id;
        f.pad = 0;
        w.ft[pl.ft_base + ex_ft] = f;
    }
    for (u32 q = 0; q < n_chunk; ++q) {
        FTile f;
        f.off = seg_off + coff[tid] + q * CAP;
        f.size = c - q * CAP < (u32)CAP ? c - q * CAP : (u32)CAP;
        f.src = dst_id;
        f.pad = 0;
        w.ft[pl.ft_base + tot_ft + ex_chunk + q] = f;
    }
    if (is_big && deep) {
        Seg s;
        s.off = seg_off + coff[tid];
        s.size = c;
        s.shift = csh;
        s.pad = (u32)(level + 1);
        deep[pl.seg_base + ex_big] = s;
    } else if (is_big) {
        Seg s;
        s.off = seg_off + coff[tid];
        s.size = c;
        s.shift = csh;
        s.pad = 0;
        w.segs[next][pl.seg_base + ex_big] = s;
        TileRef *tr = w.tiles[next] + pl.tile_base + ex_tl;
        for (u32 k = 0; k < n_tl; ++k) {
            TileRef r;
            r.seg = pl.seg_base + ex_big;
            r.k = k;
            tr[k] = r;
        }
    }
    // clean histogram rows for the new segments (they are consecutive)
    if (!deep) {
        u32 *p = (u32 *)&w.rows[next][pl.seg_base];
        const u32 words = tot_big * (u32)(sizeof(HistRow) / 4);
        for (u32 j = tid; j < words; j += WG) p[j] = 0;
    }
}

template <bool ROOT>
__global__ void __launch_bounds__(WG) k_scatter(Ws w, int level) {
    __shared__ u64 stage[TILE];
    __shared__ u32 lcnt[256], lstart[256], gbase[256], coff[256], csize[256];
    __shared__ u32 wtot[NWAVE];
    __shared__ PlanLds pl;
    const int tid = threadIdx.x;
    const int set = level & 1;
    const u32 H = w.ctl->n_heavy;
    const u64 *src = ROOT ? (H ? (const u64 *)w.B : w.in) : ((level & 1) ? (const u64 *)w.C : (const u64 *)w.B);
    u64 *dst = ROOT ? w.C : ((level & 1) ? w.B : w.C);
    const u32 dst_id = ROOT ? (u32)SRC_C : ((level & 1) ? (u32)SRC_B : (u32)SRC_C);

    u32 cur_seg = 0xFFFFFFFFu;
    Seg sg;
    sg.off = sg.size = sg.shift = sg.pad = 0;
    HistRow *row = nullptr;
    u32 n_items;  // tiles (levels >= 1) or regions (root)
    if (ROOT) {
        const u32 n_tail = w.ctl->n_tail;
        if (n_tail == 0) return;
        if (H == 0 && n_tail <= (u32)CAP) {  // the whole input is one finish tile, read where it lies
            if (blockIdx.x == 0 && tid == 0) {
                FTile f;
                f.off = 0;
                f.size = n_tail;
                f.src = SRC_IN;
                f.pad = 0;
                w.ft[0] = f;  // cap_ft >= 1
                w.ctl->nft = 1;
            }
            return;
        }
        const u64 diff = w.ctl->or_all ^ ~w.ctl->nor_all;
        sg.size = n_tail;
        sg.shift = (diff >> 56) ? 56u : 48u;
        row = &w.root[sg.shift == 56u ? 1 : 0];
        n_items = w.regions;
        u32 tot;
        const u32 v = tid < 256 ? row->cnt[tid] : 0u;
        const u32 ex = wg_excl_scan(v, wtot, tot);
        if (tid < 256) {
            coff[tid] = ex;
            csize[tid] = v;
        }
        __syncthreads();
        if (blockIdx.x == 0) {
            plan_segment(w, 0, 0u, sg.size, sg.shift, row->cnt, row->orv, row->norv, coff, dst_id, pl);
            if (tid == 0) atomicMax(&w.ctl->levels_used, 1u);
        }
    } else {
        if (w.ctl->err) return;
        n_items = w.ctl->ntile[level] < w.cap_tile ? w.ctl->ntile[level] : w.cap_tile;
    }
    for (u32 it = blockIdx.x; it < n_items; it += gridDim.x) {
        u32 tile_lo, tile_hi;  // key range of this item inside `src`, in steps of TILE
        u64 base_in;
        if (ROOT) {
            base_in = (u64)it * w.region;
            tile_lo = 0;
            tile_hi = w.rcnt[it];
        } else {
            const TileRef ref = w.tiles[set][it];
            if (ref.seg != cur_seg) {
                cur_seg = ref.seg;
                sg = w.segs[set][ref.seg];
                row = &w.rows[set][ref.seg];
                u32 tot;
                const u32 v = tid < 256 ? row->cnt[tid] : 0u;
                const u32 ex = wg_excl_scan(v, wtot, tot);
                __syncthreads();  // (coff may still be read by the previous tile's write-out)
                if (tid < 256) {
                    coff[tid] = ex;
                    csize[tid] = v;
                }
                __syncthreads();
            }
            if (ref.k == 0) plan_segment(w, level, sg.off, sg.size, sg.shift, row->cnt, row->orv, row->norv, coff, dst_id, pl);
            base_in = sg.off;
            tile_lo = ref.k * TILE;
            tile_hi = sg.size - tile_lo < (u32)TILE ? sg.size : tile_lo + TILE;
        }
        for (u32 k0 = tile_lo; k0 < tile_hi; k0 += TILE) {
            const u32 m = tile_hi - k0 < (u32)TILE ? tile_hi - k0 : (u32)TILE;
            if (tid < 256) lcnt[tid] = 0;
            __syncthreads();
            u64 k[ITEMS];
            u32 rk[ITEMS];
            const u64 *p = src + base_in + k0;
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                k[q] = j < m ? p[j] : 0ull;
            }
            if (ROOT && H == 0 && w.raw_f64) {  // (uniform: the root reads the caller's array when nothing was stripped)
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) k[q] = key_of(k[q]);
            }
            // the keys' ranks inside their bins: independent returning adds, all in flight together
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                rk[q] = j < m ? atomicAdd(&lcnt[(u32)(k[q] >> sg.shift) & 255u], 1u) : 0u;
            }
            __syncthreads();
            u32 tot;
            const u32 v = tid < 256 ? lcnt[tid] : 0u;
            const u32 ex = wg_excl_scan(v, wtot, tot);
            if (tid < 256) {
                lstart[tid] = ex;
                gbase[tid] = v ? atomicAdd(&row->cur[tid], v) : 0u;  // the bin's next free place: order is free
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                if (j < m) stage[lstart[(u32)(k[q] >> sg.shift) & 255u] + rk[q]] = k[q];
            }
            __syncthreads();
            for (u32 j = tid; j < m; j += WG) {
                const u64 key = stage[j];
                const u32 d = (u32)(key >> sg.shift) & 255u;
                const u32 in_bin = gbase[d] + (j - lstart[d]);
                // (counts that do not match the keys -- a caller overwriting them mid-sort -- must end as a
                // poisoned result, never as a store outside the bin)
                if (in_bin < csize[d]) dst[(u64)sg.off + coff[d] + in_bin] = key;
                else atomicOr(&w.ctl->err, ERR_RANGE);
            }
            __syncthreads();
        }
    }
}

// ---- the levels below the launched ones ------------------------------------------------------------------
// A segment that is still too large after the last launched level is finished by ONE workgroup through all its
// remaining levels: count, plan and scatter as above, but over the segment's own tiles in turn, with the histogram
// and the cursors in LDS and the segments it splits off on a stack of its own (depth-first; a segment's subtree
// touches nobody else's keys).  No launch per level, nothing to wait for.  Such segments are rare and small on
// the keys this library sorts (a few clustered value ranges); a large input keeps more launched levels instead.
__global__ void __launch_bounds__(WG, 4) k_deep(Ws w, int level0, Seg *arena, u32 arena_cap) {
    __shared__ u64 stage[TILE];
    __shared__ u32 hc[256];
    __shared__ u64 hor[256], hnor[256];
    __shared__ u32 lcnt[256], lstart[256], gbase[256], coff[256], csize[256], cur[256];
    __shared__ u32 wtot[NWAVE];
    __shared__ PlanLds pl;
    __shared__ u32 st_lo[MAX_LEVELS + 1], st_hi[MAX_LEVELS + 1];
    __shared__ int sp;
    const int tid = threadIdx.x;
    if (w.ctl->err) return;
    const u32 ns = w.ctl->nseg[level0] < w.cap_seg ? w.ctl->nseg[level0] : w.cap_seg;
    for (u32 si = blockIdx.x; si < ns; si += gridDim.x) {
        Seg sg = w.segs[level0 & 1][si];
        sg.pad = (u32)level0;
        if (tid == 0) sp = 0;
        for (;;) {
            // ---- one segment: sg (its level in sg.pad) ----
            const int level = (int)sg.pad;
            const u64 *src = (level & 1) ? (const u64 *)w.C : (const u64 *)w.B;
            u64 *dst = (level & 1) ? w.B : w.C;
            const u32 dst_id = (level & 1) ? (u32)SRC_B : (u32)SRC_C;
            __syncthreads();
            if (tid < 256) {
                hc[tid] = 0;
                hor[tid] = 0;
                hnor[tid] = 0;
            }
            __syncthreads();
            for (u32 k0 = 0; k0 < sg.size; k0 += TILE) {
                const u32 m = sg.size - k0 < (u32)TILE ? sg.size - k0 : (u32)TILE;
                const u64 *p = src + sg.off + k0;
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = q * WG + tid;
                    if (j < m) {
                        const u64 key = p[j];
                        const u32 d = (u32)(key >> sg.shift) & 255u;
                        atomicAdd(&hc[d], 1u);
                        atomicOr(&hor[d], key);
                        atomicOr(&hnor[d], ~key);
                    }
                }
            }
            __syncthreads();
            {
                u32 tot;
                const u32 v = tid < 256 ? hc[tid] : 0u;
                const u32 ex = wg_excl_scan(v, wtot, tot);
                if (tid < 256) {
                    coff[tid] = ex;
                    csize[tid] = v;
                    cur[tid] = 0;
                }
                __syncthreads();
                if (tot != sg.size) {  // (keys changed under the sort: poison, write nothing)
                    if (tid == 0) atomicOr(&w.ctl->err, ERR_RANGE);
                    return;
                }
            }
            plan_segment(w, level, sg.off, sg.size, sg.shift, hc, hor, hnor, coff, dst_id, pl, arena, arena_cap);
            __syncthreads();
            if (!pl.ok) return;
            for (u32 k0 = 0; k0 < sg.size; k0 += TILE) {
                const u32 m = sg.size - k0 < (u32)TILE ? sg.size - k0 : (u32)TILE;
                if (tid < 256) lcnt[tid] = 0;
                __syncthreads();
                u64 k[ITEMS];
                u32 rk[ITEMS];
                const u64 *p = src + sg.off + k0;
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = q * WG + tid;
                    k[q] = j < m ? p[j] : 0ull;
                }
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = q * WG + tid;
                    rk[q] = j < m ? atomicAdd(&lcnt[(u32)(k[q] >> sg.shift) & 255u], 1u) : 0u;
                }
                __syncthreads();
                u32 tot;
                const u32 v = tid < 256 ? lcnt[tid] : 0u;
                const u32 ex = wg_excl_scan(v, wtot, tot);
                if (tid < 256) {
                    lstart[tid] = ex;
                    gbase[tid] = cur[tid];
                    cur[tid] += v;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const u32 j = q * WG + tid;
                    if (j < m) stage[lstart[(u32)(k[q] >> sg.shift) & 255u] + rk[q]] = k[q];
                }
                __syncthreads();
                for (u32 j = tid; j < m; j += WG) {
                    const u64 key = stage[j];
                    const u32 d = (u32)(key >> sg.shift) & 255u;
                    const u32 in_bin = gbase[d] + (j - lstart[d]);
                    if (in_bin < csize[d]) dst[(u64)sg.off + coff[d] + in_bin] = key;
                    else atomicOr(&w.ctl->err, ERR_RANGE);
                }
                __syncthreads();
            }
            // ---- what it split off goes on the stack; take the next segment from there ----
            if (tid == 0) {
                if (pl.n_big) {
                    if (sp <= MAX_LEVELS) {
                        st_lo[sp] = pl.seg_base;
                        st_hi[sp] = pl.seg_base + pl.n_big;
                        ++sp;
                    } else {
                        atomicOr(&w.ctl->err, ERR_LEVEL);  // (deeper than 64-bit keys can nest: impossible)
                    }
                }
                while (sp > 0 && st_lo[sp - 1] == st_hi[sp - 1]) --sp;
            }
            __syncthreads();
            if (sp == 0) break;
            {  // (written by this workgroup's own lanes a moment ago, beside other workgroups' entries: read past L1)
                const Seg *e = arena + st_lo[sp - 1];
                sg.off = __hip_atomic_load(&e->off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sg.size = __hip_atomic_load(&e->size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sg.shift = __hip_atomic_load(&e->shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sg.pad = __hip_atomic_load(&e->pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (tid == 0) ++st_lo[sp - 1];
        }
    }
}

// ---- finish tiles ---------------------------------------------------------------------------------------
// A tile (<= CAP keys of the tail, all the keys of their value range) is ranked inside LDS.  A round spreads a
// range of the tile over BINS linear buckets between its smallest and its largest key, in place; buckets of up to
// SMALL_GROUP keys are ranked by counting; a larger bucket goes to one wave, which looks for the few distinct
// values such a bucket usually holds (a value with tens to thousands of copies and a neighbour or two) and writes
// their runs; a bucket that is not of that kind is sorted by the wave (compare-exchange network, up to WAVE_SORT
// keys) or becomes the range of a later round, bucketed over ITS OWN smallest and largest key -- clustered keys
// (a few far-apart groups, each dense) separate at the scale they have.  A round at least halves the key range of
// what it passes on, so every range ends.
constexpr int WAVE_SORT = 256;     // a bucket of up to this many keys is sorted by one wave
constexpr int WL_CAP = 128;        // ranges waiting for a round (> WAVE_SORT keys each, disjoint per depth)
constexpr int LARGE_CAP = CAP / SMALL_GROUP;
constexpr int PROBE = 4;           // distinct values a wave looks for in a large bucket

// ascending in-place sort of a[0..g) by one wave (g <= WAVE_SORT): the bitonic network in its one-direction form
// (the first step of every merge pairs i with i ^ (2k - 1)), so that the missing keys of the last power of two
// act as +infinity without being stored
__device__ __forceinline__ void wave_sort(u64 *a, u32 g, int lane) {
    u32 P = 64;
    while (P < g) P <<= 1;
    for (u32 k = 2; k <= P; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            const u32 mask = j == (k >> 1) ? k - 1 : j;
            for (u32 i = lane; i < P; i += 64) {
                const u32 l = i ^ mask;
                if (l > i && l < g) {
                    const u64 x = a[i], y = a[l];
                    if (y < x) {
                        a[i] = y;
                        a[l] = x;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// BIG (up to H_BIG heavy values): only the values themselves are held in LDS -- with their prefix sums and the gap
// counters beside them one workgroup would have a CU to itself; the prefix sums are read from memory
// (k_heavy_prefix), once per tile unless a heavy value lies inside the tile's range, and the gaps are counted there.
template <bool BIG>
__global__ void __launch_bounds__(WG, 4) k_finish(Ws w) {  // (4 waves per SIMD = two workgroups a CU: 128 VGPRs)
    constexpr u32 HM = BIG ? (u32)H_BIG : (u32)H_MAX;
    __shared__ u64 stage[CAP];
    __shared__ u32 start[BINS + 1];
    __shared__ u64 hv[HM];
    __shared__ u32 hp[BIG ? 1 : H_MAX + 1];    // copies of the heavy values below index i
    __shared__ u32 gapl[BIG ? 1 : H_MAX + 1];
    __shared__ u32 wtot[NWAVE];
    __shared__ u64 red[2 * NWAVE];
    __shared__ u32 large[2 * LARGE_CAP];
    __shared__ u32 wl[2 * WL_CAP];
    __shared__ u32 nlarge, nwl;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u32 H = w.ctl->n_heavy;
    if (w.ctl->err) return;  // a table overflowed in an earlier kernel: the tile list is not to be trusted
    const u32 nft = w.ctl->nft < w.cap_ft ? w.ctl->nft : w.cap_ft;
    if (blockIdx.x >= nft) return;
    // the first tile's keys are requested before anything else: they travel while the heavy values are set up
    // (two dependent round trips to memory, the tile's entry and its keys, were a third of a tile's time)
    u64 pre[ITEMS];
    auto request = [&](u32 t) {  // the keys of tile t (tile entries are checked before they are believed)
        const FTile f = w.ft[t];
        const u32 m = f.off < w.n ? (f.size < (u32)CAP ? f.size : (u32)CAP) : 0u;
        const u64 *src = (f.src == SRC_IN ? w.in : (f.src == SRC_B ? (const u64 *)w.B : (const u64 *)w.C)) + f.off;
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const u32 j = q * WG + tid;
            pre[q] = j < m && (u64)f.off + j < (u64)w.n ? src[j] : 0ull;
        }
        if (f.src == SRC_IN && w.raw_f64) {  // (uniform)
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) pre[q] = key_of(pre[q]);
        }
    };
    request(blockIdx.x);
    // heavy values and the prefix sums of their copies
    if (BIG) {
        for (u32 i = tid; i < H; i += WG) hv[i] = w.heavy[i];
    } else {
        u32 c[2];
        u32 s = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32 i = tid * 2 + q;
            c[q] = i < H ? w.hcnt[i] : 0u;
            if (i < H) hv[i] = w.heavy[i];
            s += c[q];
        }
        u32 tot;
        u32 ex = wg_excl_scan(s, wtot, tot);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32 i = tid * 2 + q;
            if (i <= H) hp[i] = ex;
            ex += c[q];
        }
        if (tid == 0) hp[H] = tot;
        for (u32 i = tid; i <= (u32)H_MAX; i += WG) gapl[i] = 0;
    }
    __syncthreads();
    auto heavy_below = [&](u64 key, u32 lo, u32 hi) {  // values of hv[lo..hi) that are < key, + lo
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (hv[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    for (u32 t = blockIdx.x; t < nft; t += gridDim.x) {
        const FTile f = w.ft[t];
        const u32 m = f.off < w.n ? (f.size < (u32)CAP ? f.size : (u32)CAP) : 0u;  // (as request() read it)
        u64 *out = w.A + f.off;
        const u64 room = (u64)w.n - f.off;  // a final index is < n; checked at every store
        u32 idx_lo = 0, idx_hi = 0;         // heavy values below the tile's smallest / largest key
        u32 hp_lo = 0;                      // BIG: copies of the heavy values below the tile's smallest key
        auto copies_below = [&](u32 hidx) { return BIG ? (hidx == idx_lo ? hp_lo : w.hpg[hidx]) : hp[hidx]; };
        auto count_gap = [&](u32 hidx, u32 c) {
            if (BIG) atomicAdd(&w.gap[hidx], c); else atomicAdd(&gapl[hidx], c);
        };
        // `pos` = the key's rank inside the tile; its place in A adds the copies of the heavy values below it
        auto emit = [&](u32 pos, u64 key) {
            const u32 hidx = idx_lo == idx_hi ? idx_lo : heavy_below(key, idx_lo, idx_hi);
            const u64 at = (u64)pos + copies_below(hidx);
            if (at < room) out[at] = key; else atomicOr(&w.ctl->err, ERR_RANGE);
            if (idx_lo != idx_hi) count_gap(hidx, 1u);
        };
        auto emit_run = [&](u32 pos, u64 key, u32 c, int ln, int step) {  // c copies of key from rank pos on
            const u32 hidx = idx_lo == idx_hi ? idx_lo : heavy_below(key, idx_lo, idx_hi);
            const u64 at0 = (u64)pos + copies_below(hidx);
            for (u32 i = ln; i < c; i += step) {
                if (at0 + i < room) out[at0 + i] = key; else atomicOr(&w.ctl->err, ERR_RANGE);
            }
            if (ln == 0 && idx_lo != idx_hi) count_gap(hidx, c);
        };
        if (tid == 0) {
            nlarge = 0;
            nwl = 1;
            wl[0] = 0;
            wl[1] = m;
        }
        for (u32 r = 0;; ++r) {
            __syncthreads();
            if (r >= nwl || r >= (u32)WL_CAP) break;
            const u32 s0 = wl[2 * r], g0 = wl[2 * r + 1];
            u64 k[ITEMS];
            u64 mn = EMPTY, mx = 0;
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                k[q] = j < g0 ? (r == 0 ? pre[q] : stage[s0 + j]) : 0ull;
                if (j < g0) {
                    mn = k[q] < mn ? k[q] : mn;
                    mx = k[q] > mx ? k[q] : mx;
                }
            }
            mn = wave_min(mn);
            mx = wave_max(mx);
            if (lane == 0) {
                red[wid] = mn;
                red[NWAVE + wid] = mx;
            }
            __syncthreads();
            mn = red[0];
            mx = red[NWAVE];
#pragma unroll
            for (int i = 1; i < NWAVE; ++i) {
                mn = red[i] < mn ? red[i] : mn;
                mx = red[NWAVE + i] > mx ? red[NWAVE + i] : mx;
            }
            if (r == 0) {
                idx_lo = heavy_below(mn, 0, H);
                idx_hi = heavy_below(mx, idx_lo, H);
                if (BIG) hp_lo = w.hpg[idx_lo];
            }
            if (mn == mx) {  // one value: a run
                emit_run(s0, mn, g0, tid, WG);
                continue;
            }
            // linear buckets over [mn, mx]: (key - mn) >> sh < BINS
            const u64 range = mx - mn;
            const int bl = 64 - __builtin_clzll(range);
            const int sh = bl > 12 ? bl - 12 : 0;
            for (int i = tid; i <= BINS; i += WG) start[i] = 0;
            __syncthreads();
            u32 rk[ITEMS];
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                rk[q] = j < g0 ? atomicAdd(&start[(u32)((k[q] - mn) >> sh)], 1u) : 0u;
            }
            __syncthreads();
            {
                u32 c[BINS / WG];
                u32 s = 0;
#pragma unroll
                for (int q = 0; q < BINS / WG; ++q) {
                    c[q] = start[tid * (BINS / WG) + q];
                    s += c[q];
                }
                u32 tot;
                u32 ex = wg_excl_scan(s, wtot, tot);
#pragma unroll
                for (int q = 0; q < BINS / WG; ++q) {
                    start[tid * (BINS / WG) + q] = ex;
                    ex += c[q];
                }
                if (tid == 0) start[BINS] = g0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ITEMS; ++q) {  // (in place: every key of the range is in a register)
                const u32 j = q * WG + tid;
                if (j < g0) stage[s0 + start[(u32)((k[q] - mn) >> sh)] + rk[q]] = k[q];
            }
            __syncthreads();
#pragma unroll 1
            for (int q = 0; q < ITEMS; ++q) {
                const u32 j = q * WG + tid;
                if (j >= g0) break;
                const u64 *st = stage + s0;
                const u64 key = st[j];
                const u32 d = (u32)((key - mn) >> sh);
                const u32 s = start[d], e = start[d + 1], g = e - s;
                u32 pos = j;
                if (g > (u32)SMALL_GROUP) {
                    if (j == s) {
                        const u32 li = atomicAdd(&nlarge, 1u);
                        if (li < (u32)LARGE_CAP) {
                            large[2 * li] = s0 + s;
                            large[2 * li + 1] = g;
                        } else {
                            atomicOr(&w.ctl->err, ERR_RANGE);  // (more than CAP / SMALL_GROUP such buckets: impossible)
                        }
                    }
                    continue;
                }
                if (g > 1) {
                    u32 rr = 0;
                    for (u32 i = s; i < e; i += 4) {  // (four independent reads a step)
                        const u64 o0 = st[i];
                        const u64 o1 = st[i + 1 < e ? i + 1 : i];
                        const u64 o2 = st[i + 2 < e ? i + 2 : i];
                        const u64 o3 = st[i + 3 < e ? i + 3 : i];
                        rr += (o0 < key || (o0 == key && i < j)) ? 1u : 0u;
                        rr += (i + 1 < e && (o1 < key || (o1 == key && i + 1 < j))) ? 1u : 0u;
                        rr += (i + 2 < e && (o2 < key || (o2 == key && i + 2 < j))) ? 1u : 0u;
                        rr += (i + 3 < e && (o3 < key || (o3 == key && i + 3 < j))) ? 1u : 0u;
                    }
                    pos = s + rr;
                }
                emit(s0 + pos, key);
            }
            __syncthreads();
            // this round's large buckets, a wave each
            const u32 nl = nlarge < (u32)LARGE_CAP ? nlarge : (u32)LARGE_CAP;
            for (u32 li = wid; li < nl; li += NWAVE) {
                const u32 s = large[2 * li], g = large[2 * li + 1];
                {   // one value (the usual large bucket: a score with tens to thousands of copies): a run
                    const u64 v0 = stage[s];
                    bool same = true;
                    for (u32 i = lane; i < g; i += 64) same &= stage[s + i] == v0;
                    if (__all(same)) {
                        emit_run(s, v0, g, lane, 64);
                        continue;
                    }
                }
                // the PROBE smallest distinct values and their copies -- nothing is written until they are known
                // to be all there is
                u64 cand[PROBE];
                u32 cc[PROBE];
                u32 found = 0, placed = 0;
                u64 last = 0;
                bool go = true;
#pragma unroll
                for (int it = 0; it < PROBE; ++it) {
                    cand[it] = EMPTY;
                    cc[it] = 0;
                    if (it == 2 && placed * 2 < g) go = false;  // (two values and not half the bucket: give up)
                    if (go && placed < g) {
                        u64 c0 = EMPTY;
                        for (u32 i = lane; i < g; i += 64) {
                            const u64 v = stage[s + i];
                            if ((it == 0 || v > last) && v < c0) c0 = v;
                        }
                        c0 = wave_min(c0);
                        u32 c = 0;
                        for (u32 i = lane; i < g; i += 64) c += stage[s + i] == c0 ? 1u : 0u;
                        c = wave_sum(c);
                        cand[it] = c0;
                        cc[it] = c;
                        placed += c;
                        last = c0;
                        found = it + 1;
                    }
                }
                if (placed == g) {
                    u32 at = s;
#pragma unroll
                    for (int it = 0; it < PROBE; ++it) {
                        if (it < (int)found) {
                            emit_run(at, cand[it], cc[it], lane, 64);
                            at += cc[it];
                        }
                    }
                } else if (g <= (u32)WAVE_SORT) {
                    wave_sort(stage + s, g, lane);
                    for (u32 i = lane; i < g; i += 64) emit(s + i, stage[s + i]);
                } else if (lane == 0) {
                    const u32 wi = atomicAdd(&nwl, 1u);
                    if (wi < (u32)WL_CAP) {
                        wl[2 * wi] = s;
                        wl[2 * wi + 1] = g;
                    } else {
                        atomicOr(&w.ctl->err, ERR_RANGE);  // (deeper than 64-bit keys can nest: impossible)
                    }
                }
            }
            __syncthreads();
            if (tid == 0) nlarge = 0;
        }
        // (no heavy value inside the tile's range: all its keys have the same number of heavy values below)
        if (idx_lo == idx_hi && tid == 0 && m) {
            if (BIG) atomicAdd(&w.gap[idx_lo], m); else gapl[idx_lo] += m;
        }
        if (t + gridDim.x < nft) request(t + gridDim.x);
    }
    if (BIG) return;
    __syncthreads();
    for (u32 i = tid; i <= H; i += WG)
        if (gapl[i]) atomicAdd(&w.gap[i], gapl[i]);
}

// ---- the runs of the heavy values -----------------------------------------------------------------------
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
constexpr int FILL_GRID_BIG = 2048;
template <bool BIG>
__global__ void __launch_bounds__(256) k_fill(Ws w) {
    // copies a workgroup writes between two searches of the prefix (large inputs: 8 GB of copies at 1.25e9 keys went
    // from 2.39 ms to 1.84 with 16-byte stores and four times the piece: 4.3 TB/s, what k_positions' stores reach)
    constexpr int CHUNK = BIG ? 16384 : 4096;
    constexpr u32 HM = BIG ? (u32)H_BIG : (u32)H_MAX;
    constexpr int PT = HM / 256;     // heavy values per thread
    static_assert(PT * 256 == HM, "whole heavy values per thread");
    __shared__ u64 hv[HM];
    __shared__ u32 hp[HM + 1];       // copies of heavy values below index i
    __shared__ u32 st[HM];           // final index of the first copy of heavy value i
    const int tid = threadIdx.x;
    const u32 H = w.ctl->n_heavy;
    if (H == 0 || w.ctl->err) return;
    // prefix sums over the heavy values (parallel: a serial walk of LDS by one lane costs ~100 cycles a step)
    __shared__ u32 wtot[4];
    auto scan_own = [&](const u32 (&a)[PT], u32 &total) {  // this thread's PT consecutive entries
        const int lane = tid & 63, wid = tid >> 6;
        u32 s = 0;
#pragma unroll
        for (int q = 0; q < PT; ++q) s += a[q];
        u32 inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        __syncthreads();
        if (lane == 63) wtot[wid] = inc;
        __syncthreads();
        u32 off = 0, tot = 0;
        for (int i = 0; i < 4; ++i) {
            if (i < wid) off += wtot[i];
            tot += wtot[i];
        }
        total = tot;
        return off + inc - s;
    };
    u32 c[PT], total;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        c[q] = i < H ? w.hcnt[i] : 0u;
        if (i < H) hv[i] = w.heavy[i];
    }
    u32 ex = scan_own(c, total);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        if (i <= H) hp[i] = ex;
        ex += c[q];
    }
    if (tid == 0) hp[H] = total;
    if ((u64)blockIdx.x * CHUNK >= total) return;  // (uniform)
    u32 g[PT], gtot;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        g[q] = i < H ? w.gap[i] : 0u;
    }
    u32 gex = scan_own(g, gtot);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const u32 i = tid * PT + q;
        gex += g[q];  // tail keys with at most i heavy values below them ARE below heavy value i
        if (i < H) st[i] = hp[i] + gex;
    }
    __syncthreads();
    auto run_of = [&](u64 x) {  // last i with hp[i] <= x
        u32 lo = 0, hi = H;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (hp[mid] <= x) lo = mid; else hi = mid;
        }
        return lo;
    };
    for (u64 x0 = (u64)blockIdx.x * CHUNK; x0 < total; x0 += (u64)gridDim.x * CHUNK) {
        const u64 x1 = x0 + CHUNK < total ? x0 + CHUNK : total;  // this chunk: copies x0 .. x1-1 of all heavy copies
        const u32 j0 = run_of(x0), j1 = run_of(x1 - 1);
        if (j0 == j1) {  // (the common case: runs are long)
            const u64 v = hv[j0];
            const u64 at0 = (u64)st[j0] + (x0 - hp[j0]);
            u32 len = (u32)(x1 - x0);
            if (at0 + len > w.n) {  // (a count that is wrong: no store leaves the array)
                if (tid == 0) atomicOr(&w.ctl->err, ERR_RANGE);
                len = at0 < w.n ? (u32)(w.n - at0) : 0u;
            }
            // 16 bytes a lane: a key alone at either end where the run's piece starts or ends off a 16-byte boundary
            u64 *const dst = w.A + at0;
            const u32 head = len && (((unsigned long long)dst >> 3) & 1ull) ? 1u : 0u;
            const u32 pairs = (len - head) >> 1;
            if (tid == 0 && head) dst[0] = v;
            u64x2 *const p2 = (u64x2 *)(dst + head);
            const u64x2 vv = {v, v};
            for (u32 i = tid; i < pairs; i += 256) p2[i] = vv;
            if (tid == 255 && ((len - head) & 1u)) dst[len - 1] = v;
        } else {
            for (u64 x = x0 + tid; x < x1; x += 256) {
                u32 lo = j0, hi = j1 + 1;
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if (hp[mid] <= x) lo = mid; else hi = mid;
                }
                const u64 at = (u64)st[lo] + (x - hp[lo]);
                if (at < w.n) w.A[at] = hv[lo]; else atomicOr(&w.ctl->err, ERR_RANGE);
            }
        }
    }
}

__global__ void k_force_error(Ctl *ctl) { atomicOr(&ctl->err, ERR_DEBUG); }

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
    size_t B, C, ctl, heavy, hcnt, gap, root, rcnt, segs[2], rows[2], tiles[2], ft, end;
    size_t gk, gc, ghist, glist, glcnt, hkg, hig, hpg;  // large inputs (gk = 0: not laid out)
    size_t gzero_bytes;                          // gc and ghist, cleared together
    u32 region, regions, cap_seg, cap_tile, cap_ft;
    size_t clear_from, clear_bytes;
};

Layout layout_for(u64 n) {
    Layout L;
    // keys per strip region (its tail is stored at the region's own offset): whole tiles, as few as give every
    // k_strip workgroup the GPU holds at once one region -- 814 regions of three tiles were a round of 512 and one of 302
    {
        const u64 tiles = (n + TILE - 1) / TILE;
        const u64 per = (tiles + STRIP_SLOTS - 1) / STRIP_SLOTS;
        L.region = (u32)((per ? per : 1) * TILE);
    }
    L.regions = (u32)((n + L.region - 1) / L.region);
    L.cap_seg = (u32)(n / CAP + 2);
    L.cap_tile = (u32)(n / TILE + L.cap_seg + 2);
    L.cap_ft = (u32)(3 * (n / CAP) + 3 * (u64)MAX_LEVELS * (n / CAP + 2) + 64);
    size_t p = 0;
    L.B = p; p += align256(n * sizeof(u64));
    L.C = p; p += align256(n * sizeof(u64));
    L.clear_from = p;
    L.ctl = p; p += align256(sizeof(Ctl));
    const bool big = n >= BIG_LAYOUT_N;
    const size_t hm = big ? H_BIG : H_MAX;
    L.heavy = p; p += align256(hm * sizeof(u64));
    L.hcnt = p; p += align256(hm * sizeof(u32));
    L.gap = p; p += align256((hm + 1) * sizeof(u32));
    L.root = p; p += align256(2 * sizeof(HistRow));
    L.clear_bytes = p - L.clear_from;
    L.rcnt = p; p += align256((size_t)L.regions * sizeof(u32));
    for (int i = 0; i < 2; ++i) {
        L.segs[i] = p; p += align256((size_t)L.cap_seg * sizeof(Seg));
        L.rows[i] = p; p += align256((size_t)L.cap_seg * sizeof(HistRow));
        L.tiles[i] = p; p += align256((size_t)L.cap_tile * sizeof(TileRef));
    }
    L.ft = p; p += align256((size_t)L.cap_ft * sizeof(FTile));
    L.gk = L.gc = L.ghist = L.glist = L.glcnt = L.hkg = L.hig = L.hpg = 0;
    L.gzero_bytes = 0;
    if (big) {
        L.gk = p; p += align256((size_t)G_SLOTS * sizeof(u64));
        L.gc = p; p += align256((size_t)G_SLOTS * sizeof(u32));
        L.ghist = p; p += align256((size_t)CNT_BINS * sizeof(u32));
        L.gzero_bytes = p - L.gc;
        L.glist = p; p += align256((size_t)H_BIG * sizeof(u64));
        L.glcnt = p; p += align256((size_t)H_BIG * sizeof(u32));
        L.hkg = p; p += align256((size_t)(HSET_BIG + 1) * sizeof(u64));
        L.hig = p; p += align256((size_t)(HSET_BIG + 1) * sizeof(u32));
        L.hpg = p; p += align256((size_t)(H_BIG + 1) * sizeof(u32));
    }
    L.end = p;
    return L;
}

}  // namespace

size_t mc_sort_workspace_bytes(uint64_t n) { return layout_for(n).end; }

int mc_sort_keys(const uint64_t *in, uint64_t n, uint64_t *out, void *workspace, size_t workspace_bytes,
                 uint32_t **err_word_out, int flags, hipStream_t s) {
    const bool debug_fail = (flags & MC_SORT_DEBUG_FAIL) != 0, few_repeats = (flags & MC_SORT_FEW_REPEATS) != 0;
    const Layout L = layout_for(n);
    // the larger search for heavy values: from BIG_N keys on (below it costs more than it saves), or on demand
    const bool big = !few_repeats && L.gk != 0 && !(flags & MC_SORT_SMALL_SEARCH) &&
                     (n >= BIG_N || (flags & MC_SORT_LARGE_SEARCH) != 0);
    if (workspace_bytes < L.end) {
        mc_set_error("mc_sort_keys: workspace of %zu bytes, need %zu", workspace_bytes, L.end);
        return MC_ERR_WORKSPACE;
    }
    char *base = (char *)workspace;
    Ws w;
    w.ctl = (Ctl *)(base + L.ctl);
    w.heavy = (u64 *)(base + L.heavy);
    w.hcnt = (u32 *)(base + L.hcnt);
    w.gap = (u32 *)(base + L.gap);
    w.root = (HistRow *)(base + L.root);
    w.rcnt = (u32 *)(base + L.rcnt);
    for (int i = 0; i < 2; ++i) {
        w.segs[i] = (Seg *)(base + L.segs[i]);
        w.rows[i] = (HistRow *)(base + L.rows[i]);
        w.tiles[i] = (TileRef *)(base + L.tiles[i]);
    }
    w.ft = (FTile *)(base + L.ft);
    w.in = (const u64 *)in;
    w.A = (u64 *)out;
    w.B = (u64 *)(base + L.B);
    w.C = (u64 *)(base + L.C);
    w.n = (u32)n;
    w.raw_f64 = (flags & MC_SORT_RAW_F64) ? 1u : 0u;
    w.region = L.region;
    w.regions = L.regions;
    w.cap_seg = L.cap_seg;
    w.cap_tile = L.cap_tile;
    w.cap_ft = L.cap_ft;
    w.gk = big ? (u64 *)(base + L.gk) : nullptr;
    w.gc = big ? (u32 *)(base + L.gc) : nullptr;
    w.ghist = big ? (u32 *)(base + L.ghist) : nullptr;
    w.glist = big ? (u64 *)(base + L.glist) : nullptr;
    w.glcnt = big ? (u32 *)(base + L.glcnt) : nullptr;
    w.hkg = big ? (u64 *)(base + L.hkg) : nullptr;
    w.hig = big ? (u32 *)(base + L.hig) : nullptr;
    w.hpg = big ? (u32 *)(base + L.hpg) : nullptr;
    if (debug_fail) w.cap_ft = 1;  // the first plan overflows the finish-tile table: the failure path, on demand
    if (err_word_out) *err_word_out = &w.ctl->err;

    // the small search's single workgroup clears the control tables on its way (a memset is a launch of its own:
    // 4-5 us of the 0.24 ms); the other paths start with the memset
    const bool sniff_clears = !big && !few_repeats;
    static const int env_reuse = getenv("MC_SORT_REUSE") ? atoi(getenv("MC_SORT_REUSE")) : 1;
    w.reuse_list = env_reuse && !(flags & MC_SORT_FRESH_SEARCH) ? 1u : 0u;
    w.clear_from = (u32 *)(base + L.clear_from);
    w.clear_words = sniff_clears ? (u32)(L.clear_bytes / sizeof(u32)) : 0u;
    if (!sniff_clears) HIP_TRY(hipMemsetAsync(base + L.clear_from, 0, L.clear_bytes, s));
    const u32 g_region = L.regions < 2048u ? L.regions : 2048u;
    if (big) {
        HIP_TRY(hipMemsetAsync(base + L.gk, 0xFF, (size_t)G_SLOTS * sizeof(u64), s));  // EMPTY
        HIP_TRY(hipMemsetAsync(base + L.gc, 0, L.gzero_bytes, s));
        hipLaunchKernelGGL(k_sniff_count, dim3(BIG_WGS), dim3(SNIFF_THREADS), 0, s, w);
        hipLaunchKernelGGL(k_sniff_hist, dim3(64), dim3(1024), 0, s, w);
        hipLaunchKernelGGL(k_sniff_pick, dim3(64), dim3(1024), 0, s, w);
        hipLaunchKernelGGL(k_sniff_sort, dim3(1), dim3(1024), 0, s, w);
        hipLaunchKernelGGL(k_strip<true>, dim3(g_region), dim3(WG), 0, s, w);
        hipLaunchKernelGGL(k_heavy_prefix, dim3(1), dim3(1024), 0, s, w);
    } else {
        if (!few_repeats) hipLaunchKernelGGL(k_sniff, dim3(1), dim3(SNIFF_THREADS), 0, s, w);
        hipLaunchKernelGGL(k_strip<false>, dim3(g_region), dim3(WG), 0, s, w);
    }
    hipLaunchKernelGGL(k_scatter<true>, dim3(g_region), dim3(WG), 0, s, w, 0);
    const u32 max_tiles = (u32)(n / TILE + 1);
    // Levels 1 .. deep_from - 1 are launches over all tiles of all segments; what is still too large then is
    // finished by single workgroups (k_deep).  Two launched levels resolve 24 bits of spread on top of the finish
    // tiles' 12: beyond them segments are few on these keys, and every further launched level costs two dependent
    // launches whether it finds work or not.  A large input keeps them all: there a deep segment is too much for
    // one workgroup.
    const int deep_from = n <= (1ull << 25) ? 3 : MAX_LEVELS;
    for (int level = 1; level < deep_from; ++level) {
        // deeper levels hold less and less (a bin must exceed CAP keys AND not be one repeated value to get there)
        const u32 want = level == 1 ? 2048u : level == 2 ? 1024u : 256u;
        const u32 g = max_tiles < want ? max_tiles : want;
        if (n <= (u64)CAP) break;  // one finish tile, planned by the root
        hipLaunchKernelGGL(k_count, dim3(g), dim3(WG), 0, s, w, level);
        hipLaunchKernelGGL(k_scatter<false>, dim3(g), dim3(WG), 0, s, w, level);
    }
    if (deep_from < MAX_LEVELS && n > (u64)CAP) {
        // the rows of the last launched level are dead by now: their memory is the arena of k_deep's segments
        Seg *arena = (Seg *)w.rows[(deep_from - 1) & 1];
        const u32 arena_cap = (u32)((size_t)L.cap_seg * sizeof(HistRow) / sizeof(Seg));
        const u32 g = L.cap_seg < 512u ? L.cap_seg : 512u;
        hipLaunchKernelGGL(k_deep, dim3(g), dim3(WG), 0, s, w, deep_from, arena, arena_cap);
    }
    if (debug_fail) hipLaunchKernelGGL(k_force_error, dim3(1), dim3(1), 0, s, w.ctl);
    const u32 g_ft = L.cap_ft < 1024u ? L.cap_ft : 1024u;
    if (big) {
        hipLaunchKernelGGL(k_finish<true>, dim3(g_ft), dim3(WG), 0, s, w);
        hipLaunchKernelGGL(k_fill<true>, dim3(FILL_GRID_BIG), dim3(256), 0, s, w);
    } else {
        hipLaunchKernelGGL(k_finish<false>, dim3(g_ft), dim3(WG), 0, s, w);
        // (no heavy values: no runs.  A workgroup writes 4096 copies at a time and there are at most n: no workgroups beyond)
        const u32 g_fill = (u32)(n / 4096 + 1 < 1024 ? n / 4096 + 1 : 1024);
        if (!few_repeats) hipLaunchKernelGGL(k_fill<false>, dim3(g_fill), dim3(256), 0, s, w);
    }
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

int mc_sort_forget(void *workspace, uint64_t n, hipStream_t s) {
    HIP_TRY(hipMemsetAsync((char *)workspace + layout_for(n).ctl, 0, sizeof(Ctl), s));
    return MC_OK;
}

int mc_sort_info(const void *workspace, uint64_t n, uint64_t info[8], hipStream_t s) {
    const Layout L = layout_for(n);
    Ctl c;
    HIP_TRY(hipMemcpyAsync(&c, (const char *)workspace + L.ctl, sizeof c, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    info[0] = c.err;
    info[1] = c.n_heavy;
    info[2] = c.n_tail;
    info[3] = c.levels_used;
    info[4] = c.nft;
    info[5] = c.segs_total;
    info[6] = c.list_magic == LIST_MAGIC ? c.list_age : 0;  // sorts the list of heavy values has served since its search
    info[7] = 0;
    return MC_OK;
}
