// Rank estimation (gfx950): replaces MonteCarloLib.__init__ / ml2p2rank,
// reference lib4mc/MonteCarloLib.py:10-32:
//     keys.sort(); positions = cumsum(2 ** (keys - log2 n)); bisect_right(keys, x)
//
//   sort      csrc/mc_sort.hip: heavy repeated values are counted instead of moved, the rest goes
//             through MSD radix levels and an in-LDS finish; no workgroup waits for another one.
//   scan      positions = inclusive prefix sum of exp2(key - log2 n_global), fp64,
//             reduce-then-scan over per-workgroup chunks.
//   query     one lane per test password: upper_bound over the sorted keys.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <mutex>

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

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;  // (k_positions pads its LDS tile for exactly 8)
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr int SCAN_MAX_BLOCKS = 16384;      // chunks of the scan (large inputs; 2048 up to 2^25 keys)
constexpr int SCAN_SMALL_BLOCKS = 2048;

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

// A tile whose first and last key are the same bit pattern holds ONE value (the keys are sorted): its weights are
// not read and summed but written down, (j + 1) * w after j + 1 of them -- k_weight_sums and k_positions use the
// same expression, so the carry out of such a tile IS its last position.  Sampled scores are mostly long runs of
// few values (81 % of the headline keys, 94 % at 1.25e9): the scan reads a fraction of the keys.  Which tiles of
// a chunk are such tiles is found first, one tile per lane (one round trip to memory for the whole chunk: asked
// tile by tile, that question was a chain of 300 dependent loads per workgroup at 1.25e9 keys).
constexpr int SCAN_CHUNK_TILES = 1024;  // tiles per chunk at most: n < 2^32 keys over SCAN_MAX_BLOCKS chunks

__device__ __forceinline__ bool same_bits(double a, double b) {
    return __double_as_longlong(a) == __double_as_longlong(b);
}

// first[j] = first key of tile j of [lo, hi); bit j of mixed = the tile holds more than one value
__device__ __forceinline__ void look_at_tiles(const double *__restrict__ keys, u64 lo, u64 hi, double *first,
                                              u32 *mixed) {
    const u32 ntiles = (u32)((hi - lo + SCAN_TILE - 1) / SCAN_TILE);
    for (u32 j = threadIdx.x; j < (u32)SCAN_CHUNK_TILES / 32; j += SCAN_THREADS) mixed[j] = 0;
    __syncthreads();
    for (u32 j = threadIdx.x; j < ntiles; j += SCAN_THREADS) {
        const u64 t0 = lo + (u64)j * SCAN_TILE, t1 = min(hi, t0 + SCAN_TILE);
        const double a = keys[t0], b = keys[t1 - 1];
        first[j] = a;
        if (!same_bits(a, b)) atomicOr(&mixed[j >> 5], 1u << (j & 31u));
    }
    __syncthreads();
}

__global__ void __launch_bounds__(SCAN_THREADS) k_weight_sums(const double *__restrict__ keys, u64 n, u64 chunk,
                                                               double logn, double *__restrict__ partials) {
    __shared__ double wave_tot[SCAN_THREADS / 64];
    __shared__ double first[SCAN_CHUNK_TILES];
    __shared__ u32 mixed[SCAN_CHUNK_TILES / 32];
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    double s = 0.0;
    if (lo < hi) {
        look_at_tiles(keys, lo, hi, first, mixed);
        const u32 ntiles = (u32)((hi - lo + SCAN_TILE - 1) / SCAN_TILE);
        for (u32 j = 0; j < ntiles; ++j) {  // mixed tiles: all lanes, in order
            if (!((mixed[j >> 5] >> (j & 31u)) & 1u)) continue;
            const u64 t0 = lo + (u64)j * SCAN_TILE, t1 = min(hi, t0 + SCAN_TILE);
            for (u64 i = t0 + threadIdx.x; i < t1; i += SCAN_THREADS) s += exp2(keys[i] - logn);
        }
        for (u32 j = threadIdx.x; j < ntiles; j += SCAN_THREADS) {  // one-value tiles: a lane each
            if ((mixed[j >> 5] >> (j & 31u)) & 1u) continue;
            const u64 t0 = lo + (u64)j * SCAN_TILE, t1 = min(hi, t0 + SCAN_TILE);
            s += (double)(t1 - t0) * exp2(first[j] - logn);
        }
    }
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
    // a sort whose tables overflowed (error word set, csrc/mc_sort.hip) must not pass for a result
    double carry = *err ? __longlong_as_double(0x7FF8000000000000ll) : partials[blockIdx.x];
    __shared__ double first[SCAN_CHUNK_TILES];
    __shared__ u32 mixed[SCAN_CHUNK_TILES / 32];
    if (lo >= hi) return;
    look_at_tiles(keys, lo, hi, first, mixed);  // (see k_weight_sums)
    u32 j = 0;
    for (u64 t0 = lo; t0 < hi; t0 += SCAN_TILE, ++j) {
        const u64 t1 = min(hi, t0 + SCAN_TILE);
        const bool one_value = !((mixed[j >> 5] >> (j & 31u)) & 1u);
        const double w1 = exp2(first[j] - logn);
        if (one_value) {  // (the same decision in every lane: no lane waits at a barrier below)
#pragma unroll
            for (int r = 0; r < SCAN_ITEMS; ++r) {
                const u32 e = r * SCAN_THREADS + threadIdx.x;
                if (t0 + e < t1) pos[t0 + e] = carry + (double)(e + 1) * w1;
            }
            carry += (double)(t1 - t0) * w1;
            continue;
        }
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
// One lane per test password: cnt = bisect_right(keys, x), pos = positions[cnt - 1].  A wave first narrows the range
// for all its lanes together: the answers for its smallest and its largest x bracket every lane's answer, and each
// of the two is found in rounds of 32 probes (5 rounds for 1e7 keys instead of 24 dependent reads a lane); test
// passwords usually arrive in ascending order of x (mc_order_keys), so what is left to a lane is a few steps.
// Any order of x is answered correctly (the bracket is wide then), NaN as by the plain search (cnt = n).
// RUNS: the arrays are the run index of csrc/mc_runs.hip (distinct values, their cumulative positions and counts);
// their length and the build's error word come from `info` on the device.
template <bool RUNS>
__global__ void __launch_bounds__(256) k_query(const double *__restrict__ keys, const double *__restrict__ pos, u64 n_arg,
                                               const double *__restrict__ x, u64 t, double *__restrict__ out_pos,
                                               u64 *__restrict__ out_cnt, const u64 *__restrict__ ccum,
                                               const u64 *__restrict__ info) {
    const int lane = threadIdx.x & 63;
    const u64 n = RUNS ? info[0] : n_arg;
    if (RUNS && info[1]) {  // a failed build must not pass for a result
        for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < t; q += (u64)gridDim.x * blockDim.x) {
            out_pos[q] = __longlong_as_double(0x7FF8000000000000ll);
            if (out_cnt) out_cnt[q] = ~0ull;  // (not 0: "no sample at or below x" makes the estimate 1, a guess number)
        }
        return;
    }
    for (u64 q0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) - lane; q0 < t; q0 += (u64)gridDim.x * blockDim.x) {
        const u64 q = q0 + lane;
        const bool act = q < t;
        const double v = act ? x[q] : 0.0;
        // the wave's smallest and largest x (a NaN anywhere: no bracket)
        double vmin = act ? v : __longlong_as_double(0x7FF0000000000000ll);
        double vmax = act ? v : __longlong_as_double(0xFFF0000000000000ull);
        bool odd = act && !(v == v);
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const double a2 = __shfl_xor(vmin, o), b2 = __shfl_xor(vmax, o);
            vmin = a2 < vmin ? a2 : vmin;
            vmax = b2 > vmax ? b2 : vmax;
        }
        u64 lo = 0, hi = n;
        if (__ballot(odd) == 0 && n > 4096) {
            // lanes 0..31 look for vmin's answer, lanes 32..63 for vmax's: 32 probes split the current range
            const double w = lane < 32 ? vmin : vmax;
            const int sub = lane & 31;
            u64 a = 0, b = n;  // the answer (first index with keys[i] > w) lies in [a, b]
            while (b - a > 64) {
                const u64 step = (b - a) / 33 + 1;
                const u64 p = a + step * (u64)(sub + 1) - 1;  // probes a+step-1, a+2step-1, ...
                const bool gt = p < b ? keys[p] > w : true;
                const u64 bal = __ballot(gt);
                const u32 half = lane < 32 ? (u32)bal : (u32)(bal >> 32);
                const int first = half ? __ffs((int)half) - 1 : 32;  // first probe whose key is > w
                // keys[p_first] > w: the answer is <= p_first; keys[p_{first-1}] <= w: the answer is > p_{first-1}
                const u64 nb = first < 32 ? a + step * (u64)(first + 1) - 1 : b;
                const u64 na = first > 0 ? a + step * (u64)first : a;
                a = na;
                b = nb < b ? nb : b;
            }
            lo = __shfl(a, 0);   // vmin's answer is at or after this: so is every lane's
            hi = __shfl(b, 32);  // vmax's answer is at or before this
        }
        if (act) {
            while (lo < hi) {  // first index with keys[i] > v  (bisect_right)
                const u64 mid = (lo + hi) >> 1;
                if (keys[mid] > v) hi = mid; else lo = mid + 1;
            }
            if (out_cnt) out_cnt[q] = RUNS ? (lo ? ccum[lo - 1] : 0ull) : lo;
            out_pos[q] = lo ? pos[lo - 1] : 0.0;
        }
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

u32 scan_blocks(u64 n) {
    u64 tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    // large inputs: the tail of the sorted keys (distinct values, every tile read) fills whole chunks, and with 2048
    // chunks of 300 tiles those were 120 workgroups streaming 5 MB each while the rest had finished
    const u64 most = n <= (1ull << 25) ? (u64)SCAN_SMALL_BLOCKS : (u64)SCAN_MAX_BLOCKS;
    return (u32)(tiles < most ? (tiles ? tiles : 1) : most);
}

std::atomic<int> g_debug_fail{0};
std::atomic<int> g_search_mode{0};  // 0 = by size, 1 = the large search for heavy values wherever it is laid out, 2 = never

}  // namespace

extern "C" size_t mc_rank_workspace_bytes(uint64_t n) {
    return align256(mc_sort_workspace_bytes(n)) + align256(SCAN_MAX_BLOCKS * sizeof(double));
}

extern "C" void mc_rank_debug_fail(int on) { g_debug_fail.store(on ? 1 : 0); }

extern "C" void mc_rank_debug_search(int mode) { g_search_mode.store(mode == 1 || mode == 2 ? mode : 0); }

namespace {
int sort_flags() {
    const int mode = g_search_mode.load();
    return (g_debug_fail.load() ? MC_SORT_DEBUG_FAIL : 0) |
           (mode == 1 ? MC_SORT_LARGE_SEARCH : mode == 2 ? MC_SORT_SMALL_SEARCH : 0);
}
}  // namespace

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
    const size_t sort_bytes = align256(mc_sort_workspace_bytes(n));
    double *partials = (double *)((char *)workspace + sort_bytes);
    uint32_t *err = nullptr;
    {
        int rc = mc_sort_keys((const uint64_t *)keys_in, n, (uint64_t *)keys_sorted, workspace, sort_bytes, &err,
                              sort_flags(), s);
        if (rc != MC_OK) return rc;
    }
    // positions
    const double logn = std::log2((double)n_global);
    const u32 sb = scan_blocks(n);
    const u64 stiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const u64 schunk = ((stiles + sb - 1) / sb) * SCAN_TILE;
    if (schunk / SCAN_TILE > (u64)SCAN_CHUNK_TILES) {  // (cannot happen below 2^32 keys)
        mc_set_error("mc_rank_build: %llu tiles per scan chunk", (unsigned long long)(schunk / SCAN_TILE));
        return MC_ERR_ARG;
    }
    hipLaunchKernelGGL(k_weight_sums, dim3(sb), dim3(SCAN_THREADS), 0, s, (const double *)keys_sorted, (u64)n, schunk,
                       logn, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(SCAN_THREADS), 0, s, partials, sb);
    hipLaunchKernelGGL(k_positions, dim3(sb), dim3(SCAN_THREADS), 0, s, (const double *)keys_sorted, (u64)n, schunk,
                       logn, (const double *)partials, (const u32 *)err, positions);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_workspace_forget(void *workspace, uint64_t n, void *stream) {
    if (!workspace || n == 0 || n >= (1ull << 32)) {
        mc_set_error("mc_workspace_forget: null workspace or a key count outside 1 .. 2^32 - 1");
        return MC_ERR_ARG;
    }
    return mc_sort_forget(workspace, n, (hipStream_t)stream);
}

extern "C" int mc_rank_check(const void *workspace, uint64_t n, void *stream, uint64_t info[8]) {
    uint64_t local[8] = {0};
    if (!workspace || n == 0) {
        if (info) for (int i = 0; i < 8; ++i) info[i] = 0;
        return MC_OK;
    }
    int rc = mc_sort_info(workspace, n, local, (hipStream_t)stream);
    if (rc != MC_OK) return rc;
    if (info) for (int i = 0; i < 8; ++i) info[i] = local[i];
    if (local[0]) {
        mc_set_error("mc_rank_build failed on the device (error word 0x%llx: a sort table overflowed or a count did "
                     "not match its keys -- were the keys overwritten while the build ran?); positions are NaN",
                     (unsigned long long)local[0]);
        return MC_ERR_WORKSPACE;
    }
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
    // (control block, heavy-value counters, root histograms) on `stream` before its first kernel, and the
    // sort initialises every other table entry before it is read
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
    hipLaunchKernelGGL(k_query<false>, dim3(g), dim3(256), 0, (hipStream_t)stream, keys_sorted, positions, (u64)n, x,
                       (u64)t, out_pos, (u64 *)out_cnt, (const u64 *)nullptr, (const u64 *)nullptr);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_rank_runs_query(const double *vals, const uint64_t *cum_count, const double *cum_pos,
                                  const uint64_t *info, const double *x, uint64_t t, double *out_pos,
                                  uint64_t *out_cnt, void *stream) {
    if (t == 0) return MC_OK;
    if (!x || !out_pos || !vals || !cum_count || !cum_pos || !info) {
        mc_set_error("mc_rank_runs_query: bad argument");
        return MC_ERR_ARG;
    }
    u64 want = (t + 255) / 256;
    u32 g = (u32)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_query<true>, dim3(g), dim3(256), 0, (hipStream_t)stream, vals, cum_pos, (u64)0, x, (u64)t,
                       out_pos, (u64 *)out_cnt, (const u64 *)cum_count, (const u64 *)info);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}
