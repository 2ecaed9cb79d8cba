// Device residency + the sampling and scoring kernels (gfx950, wave64).
//
// Sampling (replaces MonteCarlo.sample -> sample1 -> pick_expand, reference
// lib4mc/MonteCarloParent.py:33-46, nwords_simulator.py:92-116, lib4mc/ProbLib.py:31-43):
// one lane per sampled string, persistent: a lane that finishes a string immediately
// takes the next sample id of its wave's range, so lanes never wait for the longest string
// of their wave; finished keys leave through an LDS ring, a 128-byte line at a time.  Every step of a lane is one draw = one 16-byte record
// of the bucketed draw table (buckets that straddle a threshold: + threshold(s) + 16 B).
// The accumulation `prob += -log2 p` is done in fp64 in draw order, so ml2p is
// bit-identical to the reference for the same uniforms.
//
// Scoring (replaces calc_ml2p for single-character alphabets, nwords_simulator.py:36-90,
// backwords_simulator.py:28-49): one lane per test password walking the same automaton,
// looking transitions up in the (row,symbol) hash and following parent links to back off.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

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

struct alignas(16) StateRec {
    double total;
    uint32_t ent_off, len, guide_off, gbits, parent, ctx_len;
};
struct alignas(16) Payload {
    double ml2p;
    uint32_t next, sym;
};
struct alignas(16) Slot {
    unsigned long long key;
    uint32_t val, pad;
};
static_assert(sizeof(StateRec) == sizeof(mc_state_rec) && sizeof(StateRec) == 32, "state record layout");
static_assert(sizeof(Payload) == sizeof(mc_payload) && sizeof(Payload) == 16, "payload layout");
static_assert(sizeof(Slot) == sizeof(mc_slot) && sizeof(Slot) == 16, "slot layout");

struct DevModel {
    const StateRec *state;
    const mc_draw_rec *draw;
    const mc_draw_rec *spay;
    const double *rcum;
    const Payload *payload;
    const Slot *trans;
    const Slot *ctxhash;
    const uint32_t *sym_chars;
    const Slot *wordhash;
    const unsigned long long *wordck;  // the two-choice table of the same map (mc_internal.h), or null
    const uint32_t *sym_text_off, *sym_text;
    const Payload *dense;
    unsigned long long *status;
    unsigned long long trans_mask, ctx_mask;
    uint32_t start_state, root, end_sym, min_len, max_len, flags, n_ctx, n_sym;
    uint32_t start_draw, start_gbits;
    unsigned long long word_mask, wordck_mask;
    uint32_t max_word_len, start_sym, end_text_off, end_text_len;
    // MC_KIND_BPE
    const uint32_t *seg_off, *seg_row, *seg_len;
    const Slot *labhash;
    const uint32_t *lab_off, *lab_units, *lab_cand_off, *lab_cand, *np_len, *np_off, *np_cand;
    unsigned long long lab_mask;
    uint32_t n_np, kind;
};

__device__ __forceinline__ unsigned long long *status_slot(const DevModel &m) {
    return m.status + (size_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) & (MC_STATUS_SLOTS - 1)) * MC_STATUS_STRIDE;
}

struct SampleArgs {
    unsigned long long first_id, count, seed;
    const double *uniforms;  // injected mode
    double *out_ml2p;
    uint32_t *out_len, *out_draws, *out_syms;
    uint32_t max_draws, sym_stride;
    uint32_t hot_recs;  // draw-table records cached in LDS by k_sample
    float skew;         // k_sample: workgroup i of g owns the ids between count * f(i / g) and count * f((i + 1) / g),
                        // f(x) = (1 + skew) x - skew x^2: the first workgroups' ranges are (1 + skew) x the mean, the last (1 - skew) x
    // mc_sample_carry: the first carry_groups workgroups of k_sample's grid sample nothing; they copy carry_n16
    // 16-byte words from carry_src to carry_dst (pinned host memory as a rule) while the others sample
    const uint4 *carry_src;
    uint4 *carry_dst;
    unsigned long long carry_n16;
    uint32_t carry_groups;
};

// The copy mc_sample_carry takes along: `groups` workgroups of 256 lanes move n16 16-byte words, eight loads of a
// lane in flight before their stores: ~15 GB/s a workgroup.  The pace is meant: 8 MB pushed over PCIe at the link's
// full rate (8 workgroups, 0.18 ms) slowed the sampling beside it from 0.378 to 0.472 ms, at 3 workgroups (0.2 ms) to
// 0.41 ms -- and 0.03 ms of that is the price of ANY workgroup that outlives the first generation of the grid, even
// one that only sleeps, wherever it sits in the grid (scripts/carry_probe.py)
__device__ __forceinline__ void carry_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, unsigned long long n16,
                                           unsigned int group, unsigned int groups) {
    constexpr int U = 8;
    const unsigned long long stride = (unsigned long long)groups * 256u;
    unsigned long long i = (unsigned long long)group * 256u + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

constexpr uint32_t PH_M0 = 0xD2511F53u, PH_M1 = 0xCD9E8D57u, PH_W0 = 0x9E3779B9u, PH_W1 = 0xBB67AE85u;

// Philox4x32-10; same bits as oracle/philox.py
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)PH_M0 * c0, p1 = (unsigned long long)PH_M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += PH_W0;
        k1 += PH_W1;
    }
    o0 = c0;
    o1 = c1;
    o2 = c2;
    o3 = c3;
}

struct alignas(16) DrawRec {
    uint32_t w0, w1, w2, w3;
};
static_assert(sizeof(DrawRec) == sizeof(mc_draw_rec), "draw record layout");

__device__ __forceinline__ double rec_f64(const DrawRec &q) {
    return __longlong_as_double((long long)(((unsigned long long)q.w1 << 32) | q.w0));
}

// One lane walks NC sampled strings at a time (independent chains: their table reads are in
// flight together).  A loop iteration produces, per chain, one Philox block (two 53-bit uniforms:
// draws 2j and 2j+1 of that chain's sample) in uniform control flow and then takes up to two draws
// per chain; a chain whose string ends on the first draw idles for the second (its next string
// starts with a Philox block of its own).  A draw reads ONE 16-byte record of the draw table
// (mc_draw_rec, include/mi355mc.h); only buckets that straddle a threshold read more.  Finished
// strings are written at the end of the iteration, so no load waits behind a store.
constexpr unsigned int SAMPLE_WAVES = 4;  // k_sample's workgroup: 256 lanes
constexpr unsigned int LINE_KEYS = 16;    // fp64 keys in 128 bytes
constexpr unsigned int RING_LINES = 8;    // lines a wave keeps open
constexpr unsigned int CHUNK_LINES = 4;   // lines (64 ids) a wave takes from its workgroup's cursor at a time
constexpr unsigned int RING_CHUNKS = 4;   // chunk starts a wave remembers (> RING_LINES / CHUNK_LINES + 1)

// CHARS: how a symbol's length is known -- 0 every symbol is one character, 1 read from sym_chars, 2 carried by the
// draw record (MC_DRAW_LEN_SHIFT)
// EXTRAS: the strings, their lengths or their draw counts are wanted (out_syms / out_len / out_draws); without
// them -- the estimator's own call: scores only -- the kernel carries no state for them (it is bound by the
// instructions it issues: DESIGN.md 3.1)
template <bool INJECT, int CHARS, int NC, bool EXTRAS = true>
__global__ void __launch_bounds__(256) k_sample(DevModel m, SampleArgs a) {
    // the head of the draw table (the rows with the most visits per bucket: mc_table.cpp places them
    // first) lives in LDS: a.hot_recs records, loaded once per workgroup
    extern __shared__ __align__(16) unsigned char hot_raw[];
    DrawRec *hot = (DrawRec *)hot_raw;
    if (blockIdx.x < a.carry_groups) {  // (whole workgroups, before any barrier: mc_sample_carry's copy)
        carry_copy(a.carry_src, a.carry_dst, a.carry_n16, blockIdx.x, a.carry_groups);
        return;
    }
    const unsigned int bid = blockIdx.x - a.carry_groups;
    const unsigned int gdim = gridDim.x - a.carry_groups;  // the sampling grid
    __shared__ unsigned int blk_cnt[5];
    __shared__ unsigned int cursor;
    // finished keys wait here until their 128-byte line is whole (see "finished strings" below)
    __shared__ double ring_keys[SAMPLE_WAVES][RING_LINES * LINE_KEYS];
    __shared__ unsigned int ring_bits[SAMPLE_WAVES][RING_LINES];
    __shared__ unsigned int ring_chunk[SAMPLE_WAVES][RING_CHUNKS];
    for (uint32_t i = threadIdx.x; i < a.hot_recs; i += blockDim.x) hot[i] = ((const DrawRec *)m.draw)[i];
    // a workgroup owns `per` consecutive sample ids (a multiple of 16, so that every chunk a wave takes starts on
    // a line of out_ml2p when that array is 128-byte aligned); its waves take them 64 at a time from a cursor
    unsigned long long b0, b1;
    if (a.skew > 0.0f) {
        auto bound = [&](unsigned int i) -> unsigned long long {
            if (i >= gdim) return a.count;
            const double x = (double)i / (double)gdim;
            const unsigned long long v = (unsigned long long)((double)a.count * ((1.0 + (double)a.skew) * x - (double)a.skew * x * x));
            return min(a.count, v & ~(unsigned long long)(LINE_KEYS - 1));
        };
        b0 = bound(bid);
        b1 = bound(bid + 1);
    } else {
        const unsigned long long per = ((a.count + gdim - 1) / gdim + (LINE_KEYS - 1)) & ~(unsigned long long)(LINE_KEYS - 1);
        b0 = min(a.count, (unsigned long long)bid * per);
        b1 = min(a.count, b0 + per);
    }
    const unsigned int nblk = (unsigned int)(b1 - b0);
    const unsigned int wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    double *const rkeys = ring_keys[wid];
    unsigned int *const rbits = ring_bits[wid];
    unsigned int *const rchunk = ring_chunk[wid];
    if (lane < RING_LINES) rbits[lane] = 0;
    // A wave numbers the ids it takes 0, 1, 2, ... (its "slots"): slot v is id (start of chunk v / 64) + v % 64 of
    // the workgroup's range, line v / 16 of the wave.  The chunk being handed out and the one after it are held
    // in registers (a lane's next id is on the critical path of its next draw: no LDS round trip there);
    // rchunk[] remembers the starts of the last RING_CHUNKS chunks for the lines still open.  All wave-uniform:
    unsigned int w_next = 0;      // next slot to hand out
    unsigned int w_cs = 0;        // first slot of the current chunk
    unsigned int w_c0, w_c1;      // start ids of the current chunk and of the next one
    unsigned int w_head = 0;      // first line of the wave that has not been written out
    unsigned int w_done = 0;      // bit j: line w_head + j has been written out
    if (threadIdx.x == 0) cursor = 0;
    if (threadIdx.x < 5) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    auto take_chunk = [&](unsigned int index) {  // (wave-uniform) the next 64 ids of the workgroup's range
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&cursor, CHUNK_LINES * LINE_KEYS);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        base = base < nblk ? base : nblk;
        if (lane == 0) rchunk[index % RING_CHUNKS] = base;
        return base;
    };
    w_c0 = take_chunk(0);
    w_c1 = take_chunk(1);
    auto line_base = [&](unsigned int line) {  // first id (in the workgroup's range) of an open line of this wave
        return rchunk[(line / CHUNK_LINES) % RING_CHUNKS] + (line % CHUNK_LINES) * LINE_KEYS;
    };
    // hands the wave's next slots to the lanes of `want`: slot and id (wave-uniform control flow)
    auto claim = [&](bool want, unsigned int &slot, unsigned int &id) -> bool {
        constexpr unsigned int CHUNK = CHUNK_LINES * LINE_KEYS;
        const unsigned long long f = __builtin_amdgcn_ballot_w64(want);
        const unsigned int cnt = (unsigned int)__popcll(f);
        if (cnt == 0) return false;
        const unsigned int rank =
            __builtin_amdgcn_mbcnt_hi((unsigned int)(f >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)f, 0u));
        const unsigned int mine = w_next + rank, in_chunk = mine - w_cs;
        const unsigned int at = in_chunk < CHUNK ? w_c0 + in_chunk : w_c1 + (in_chunk - CHUNK);
        if (want) {
            slot = mine;
            id = at;
        }
        w_next += cnt;
        // the ring holds RING_LINES lines from w_head on: a line that a slow string keeps open is written out
        // as far as it is complete and its stragglers go to memory directly, so nobody ever waits
        while ((w_next - 1) / LINE_KEYS >= w_head + RING_LINES) {
            const unsigned int e = w_head % RING_LINES;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned int bits = rbits[e];
            if (lane < LINE_KEYS && ((bits >> lane) & 1u))
                a.out_ml2p[b0 + line_base(w_head) + lane] = rkeys[e * LINE_KEYS + lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) rbits[e] = 0;
            ++w_head;
            w_done >>= 1;
            while (w_done & 1u) {
                w_done >>= 1;
                ++w_head;
            }
        }
        if (w_next >= w_cs + CHUNK) {
            // (the table entry the new chunk takes belonged to the chunk RING_CHUNKS before it, whose lines are
            // all below w_head: every slot handed out lies within RING_LINES lines of w_head)
            w_cs += CHUNK;
            w_c0 = w_c1;
            w_c1 = take_chunk(w_cs / CHUNK + 1);
        }
        return want && at < nblk;
    };

    unsigned int c_draws = 0, c_probes = 0, c_restart = 0, c_fail = 0, c_indirect = 0;
    unsigned int local[NC], slot[NC];
    uint32_t *sym_row[NC];  // where this chain's string goes (null: strings not wanted)
    uint32_t goff[NC], gbits[NC], plen[NC], nsym[NC], draw[NC];
    double acc[NC];
    bool live[NC];
    bool any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        local[c] = slot[c] = 0;
        live[c] = claim(true, slot[c], local[c]);
        any |= live[c];
        sym_row[c] = EXTRAS && a.out_syms ? a.out_syms + (b0 + local[c]) * a.sym_stride : nullptr;
        goff[c] = m.start_draw;
        gbits[c] = m.start_gbits;
        plen[c] = nsym[c] = draw[c] = 0;
        acc[c] = 0.0;
    }
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    const bool keep_len = (m.flags & MC_FLAG_KEEP_LEN_ON_RESTART) != 0;
    const DrawRec *__restrict__ dtab = (const DrawRec *)m.draw;
    const DrawRec *__restrict__ spay = (const DrawRec *)m.spay;
    const double *__restrict__ rcum = m.rcum;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);

    while (__builtin_amdgcn_ballot_w64(any) != 0) {  // (the whole wave stays: its lanes write lines out together)
        // ---- two uniforms per chain (draw[c] is even here) ---------------------------------
        // (the 53-bit uniform of a draw is (uh >> 5) << 26 | ul >> 6; its bucket needs the top gbits <= 20 bits only, which
        // are uh's: the 64-bit value is put together where a threshold is compared, one draw in eighty)
        uint32_t uh[NC][2], ul[NC][2];
        double rr[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long i = b0 + local[c];
            if (INJECT) {
                uh[c][0] = uh[c][1] = ul[c][0] = ul[c][1] = 0;
                rr[c][0] = rr[c][1] = 0.0;
                if (live[c]) {
                    const double *u = a.uniforms + i * a.max_draws + draw[c];
                    rr[c][0] = u[0];
                    if (draw[c] + 1 < a.max_draws) rr[c][1] = u[1];
                }
            } else {
                const unsigned long long sid = a.first_id + i;
                uint32_t o0, o1, o2, o3;
                philox4x32_10((uint32_t)sid, (uint32_t)(sid >> 32), draw[c] >> 1, 0u, k0, k1, o0, o1, o2, o3);
                uh[c][0] = o0;
                ul[c][0] = o1;
                uh[c][1] = o2;
                ul[c][1] = o3;
                rr[c][0] = rr[c][1] = 0.0;
            }
        }
        bool fresh[NC];
        double res_ml2p[NC];
        uint32_t res_len[NC] = {}, res_draws[NC] = {};
#pragma unroll
        for (int c = 0; c < NC; ++c) fresh[c] = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            DrawRec q[NC];
            bool go[NC], ind[NC];
            uint32_t idx[NC];
            // ---- phase 1: every chain's bucket record -----------------------------------------
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                go[c] = live[c] && !fresh[c];
                q[c] = DrawRec{0, 0, 0, 0};
                if (go[c]) {
                    uint32_t k;
                    if (INJECT) {
                        const double kk = rr[c][h] * (double)(1u << gbits[c]);
                        const uint32_t kmax = (1u << gbits[c]) - 1u;
                        k = kk >= (double)kmax ? kmax : (kk > 0.0 ? (uint32_t)kk : 0u);
                    } else {
                        k = (uh[c][h] >> 1) >> (31u - gbits[c]);  // = the uniform >> (53 - gbits), gbits <= 20 (mc_table.cpp)
                    }
                    const uint32_t at = goff[c] + k;
                    if (at < a.hot_recs) q[c] = hot[at]; else q[c] = dtab[at];
                }
            }
            // ---- phase 2: buckets that straddle thresholds: same index as
            //      bisect_right(cum, total * r), decided in r's own domain ----------------------
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint32_t type = MC_DRAW_TYPE(q[c].w3);
                ind[c] = go[c] && type != MC_DRAW_DIRECT;
                idx[c] = q[c].w2;
                if (ind[c]) {
                    const unsigned long long m53 = ((unsigned long long)(uh[c][h] >> 5) << 26) | (unsigned long long)(ul[c][h] >> 6);
                    const double r = INJECT ? rr[c][h] : (double)m53 * 0x1.0p-53;
                    if (type == MC_DRAW_ONE) {
                        idx[c] += (r >= rec_f64(q[c])) ? 1u : 0u;
                        ++c_probes;
                    } else {
                        uint32_t lo = 0, hi = q[c].w0;
                        const double *T = rcum + q[c].w2;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (T[mid] > r) hi = mid; else lo = mid + 1;
                            ++c_probes;
                        }
                        idx[c] += lo;
                    }
                    ++c_indirect;
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (ind[c]) q[c] = spay[idx[c]];
            // ---- phase 3: sample1's control flow (nwords_simulator.py:100-115) -----------------
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!go[c]) continue;
                acc[c] += rec_f64(q[c]);
                ++draw[c];  // (counted into c_draws when the string is done: one add a string, not one a draw)
                bool done = false, ok = false, restart = false;
                if (q[c].w3 & MC_DRAW_END) {
                    if (plen[c] >= m.min_len) done = ok = true; else restart = true;
                } else {
                    const uint32_t s = MC_DRAW_SYM(q[c].w3);
                    if (EXTRAS) {
                        if (sym_row[c] && nsym[c] < a.sym_stride) sym_row[c][nsym[c]] = s;
                        ++nsym[c];
                    }
                    if (CHARS == 2) {
                        const uint32_t off = q[c].w2 & MC_DRAW_OFF_MASK;
                        plen[c] += q[c].w2 >> MC_DRAW_LEN_SHIFT;
                        goff[c] = off == MC_DRAW_OFF_MASK ? MC_NONE : off;
                    } else {
                        plen[c] += CHARS ? m.sym_chars[s] : 1u;
                        goff[c] = q[c].w2;
                    }
                    gbits[c] = MC_DRAW_GBITS(q[c].w3);
                    if (plen[c] >= m.max_len) restart = true;
                }
                if (restart) {
                    goff[c] = m.start_draw;
                    gbits[c] = m.start_gbits;
                    acc[c] = 0.0;
                    nsym[c] = 0;
                    if (!keep_len) plen[c] = 0;
                    ++c_restart;
                }
                if (!done && (draw[c] >= a.max_draws || goff[c] == MC_NONE)) {
                    done = true;  // ok stays false: the reference would loop on / crash here
                    ++c_fail;
                }
                if (done) {
                    fresh[c] = true;
                    c_draws += draw[c];
                    res_ml2p[c] = ok ? acc[c] : nan;
                    if (EXTRAS) {
                        res_len[c] = nsym[c];
                        res_draws[c] = draw[c];
                    }
                    goff[c] = m.start_draw;
                    gbits[c] = m.start_gbits;
                    acc[c] = 0.0;
                    plen[c] = nsym[c] = draw[c] = 0;
                }
            }
        }
        // ---- finished strings: write, claim the next id ------------------------------------------
        // A key does not go to memory by itself (an 8-byte store into a line whose other keys arrive hundreds of
        // cycles later was measured as 26 bytes written to HBM per key: the gathers push the half-filled line
        // out of L2 first).  It waits in the wave's ring until the 16 keys of its line are there, and 16 lanes
        // store the line at once.
        any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bool whole = false;
            unsigned int line = 0;
            if (fresh[c]) {
                const unsigned long long i = b0 + local[c];
                line = slot[c] / LINE_KEYS;
                if (line < w_head) {
                    a.out_ml2p[i] = res_ml2p[c];  // (its line was written out without it: a straggler)
                } else {
                    const unsigned int e = line % RING_LINES, k = slot[c] % LINE_KEYS, bit = 1u << k;
                    rkeys[e * LINE_KEYS + k] = res_ml2p[c];
                    const unsigned int first = local[c] - k;  // (a line's ids: `first` .. nblk - 1 at most)
                    const unsigned int full = nblk - first >= LINE_KEYS ? (1u << LINE_KEYS) - 1u : (1u << (nblk - first)) - 1u;
                    whole = ((atomicOr(&rbits[e], bit) | bit) == full);
                }
                if (EXTRAS) {
                    if (a.out_len) a.out_len[i] = res_len[c];
                    if (a.out_draws) a.out_draws[i] = res_draws[c];
                }
            }
            const bool got = claim(fresh[c], slot[c], local[c]);  // (before the ring's answer is waited for)
            unsigned long long wm = __builtin_amdgcn_ballot_w64(whole);
            if (wm) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                while (wm) {
                    const int src = __builtin_ctzll(wm);
                    wm &= wm - 1;
                    const unsigned int ln = (unsigned int)__builtin_amdgcn_readlane((int)line, src);
                    if (ln < w_head) continue;  // (the claim above has just written it out, whole)
                    const unsigned int e = ln % RING_LINES, first = line_base(ln);
                    if (lane < LINE_KEYS && first + lane < nblk)
                        a.out_ml2p[b0 + first + lane] = rkeys[e * LINE_KEYS + lane];
                    if (lane == 0) rbits[e] = 0;
                    w_done |= 1u << (ln - w_head);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                while (w_done & 1u) {
                    w_done >>= 1;
                    ++w_head;
                }
            }
            if (fresh[c]) {
                live[c] = got;
                if (EXTRAS && a.out_syms) sym_row[c] = a.out_syms + (b0 + local[c]) * a.sym_stride;
            }
            any |= live[c];
        }
    }
    atomicAdd(&blk_cnt[0], c_draws);
    atomicAdd(&blk_cnt[1], c_probes);
    atomicAdd(&blk_cnt[2], c_restart);
    atomicAdd(&blk_cnt[3], c_fail);
    atomicAdd(&blk_cnt[4], c_indirect);
    __syncthreads();
    if (threadIdx.x < 5 && blk_cnt[threadIdx.x]) {
        const int word = threadIdx.x < 4 ? MC_CNT_DRAWS + threadIdx.x : MC_CNT_INDIRECT;
        atomicAdd(&status_slot(m)[word], (unsigned long long)blk_cnt[threadIdx.x]);
    }
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t find_entry(const DevModel &m, uint32_t state, uint32_t s, unsigned int &c_probe) {
    uint32_t row = state;
    while (row != MC_NONE) {  // longest suffix row that lists s (backwords_simulator.py:31-38)
        const unsigned long long key = ((unsigned long long)row << 32) | s;
        unsigned long long q = mix64(key + 0x9E3779B97F4A7C15ull) & m.trans_mask;
        for (;;) {
            const Slot sl = m.trans[q];
            ++c_probe;
            if (sl.key == key) return sl.val;
            if (sl.key == ~0ull) break;
            q = (q + 1) & m.trans_mask;
        }
        row = m.state[row].parent;
    }
    return MC_NONE;
}

// find_entry with the row's parent requested beside the row's first probe (one round trip a backoff level, not two)
__device__ __forceinline__ uint32_t find_entry_ahead(const DevModel &m, uint32_t state, uint32_t s, unsigned int &c_probe) {
    uint32_t row = state;
    while (row != MC_NONE) {
        const unsigned long long key = ((unsigned long long)row << 32) | s;
        unsigned long long q = mix64(key + 0x9E3779B97F4A7C15ull) & m.trans_mask;
        const uint32_t parent = m.state[row].parent;
        for (;;) {
            const Slot sl = m.trans[q];
            ++c_probe;
            if (sl.key == key) return sl.val;
            if (sl.key == ~0ull) break;
            q = (q + 1) & m.trans_mask;
        }
        row = parent;
    }
    return MC_NONE;
}

// DENSE: the model has the resolved (row, symbol) table, a step is one 16-byte gather; otherwise
// the (row,symbol) hash is probed along the backoff chain.  A lane walks NP passwords at a time
// (independent chains: the kernel is bound by gather latency, not by issue).
template <bool DENSE, int NP>
__global__ void __launch_bounds__(256) k_score(DevModel m, const uint32_t *__restrict__ syms,
                                               const unsigned long long *__restrict__ off, unsigned long long count,
                                               double *__restrict__ out) {
    unsigned int c_probe = 0, c_step = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t0 < count;
         t0 += stride * NP) {
        unsigned long long j[NP], b[NP];
        uint32_t state[NP], s_next[NP];
        double acc[NP];
        bool run[NP], ok[NP];
        bool any = false;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const unsigned long long t = t0 + c * stride;
            run[c] = t < count;
            ok[c] = true;
            j[c] = b[c] = 0;
            if (run[c]) {
                j[c] = off[t];
                b[c] = off[t + 1];
            }
            state[c] = m.start_state;
            acc[c] = 0.0;
            s_next[c] = (run[c] && j[c] < b[c]) ? syms[j[c]] : m.end_sym;
            any |= run[c];
        }
        while (any) {
            Payload p[NP];
            bool step[NP];
            uint32_t s[NP];
            // symbols of this step; the next ones are requested before this step's gathers, so that
            // they are all in flight together
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                s[c] = s_next[c];
                s_next[c] = (run[c] && j[c] + 1 < b[c]) ? syms[j[c] + 1] : m.end_sym;
                step[c] = run[c];
                if (run[c] && (s[c] >= m.n_sym || state[c] == MC_NONE)) {
                    ok[c] = false;
                    run[c] = step[c] = false;
                }
            }
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                if (!step[c]) continue;
                if (DENSE) {
                    p[c] = m.dense[(size_t)state[c] * m.n_sym + s[c]];
                    ++c_probe;
                } else {
                    const uint32_t e = find_entry(m, state[c], s[c], c_probe);
                    p[c] = e == MC_NONE ? Payload{0.0, MC_NONE, MC_NONE} : m.payload[e];
                    if (e != MC_NONE) p[c].sym = e;
                }
            }
            any = false;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                if (step[c]) {
                    if (p[c].sym == MC_NONE) {  // no suffix row lists the symbol
                        ok[c] = false;
                        run[c] = false;
                    } else {
                        acc[c] += p[c].ml2p;
                        state[c] = p[c].next;
                        ++c_step;
                        if (j[c] == b[c]) run[c] = false;  // that was the end symbol
                        ++j[c];
                    }
                }
                any |= run[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const unsigned long long t = t0 + c * stride;
            if (t < count) out[t] = (ok[c] && acc[c] < 1022.0) ? acc[c] : 1022.0;
        }
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        c_probe += __shfl_down(c_probe, o);
        c_step += __shfl_down(c_step, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_PROBES], (unsigned long long)c_probe);
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_STEPS], (unsigned long long)c_step);
    }
}

// ---- calc_ml2p for any alphabet: the DFS of _structures (nwords_simulator.py:36-60) with an
// explicit stack, one lane per password.  A frame is one activation of _structures: the offset its
// `pwd` argument starts at, the next split length to try, the automaton state (the row for the
// container so far) and the running sum of -log2 p.  sum([..]) in the reference adds left to
// right from 0, which is the same chain of fp64 additions as the running sum.
struct alignas(16) Frame {
    double acc;
    uint32_t state;
    uint16_t pos, idx;
};

struct WordsArgs {
    const uint32_t *cps;
    const unsigned long long *off;
    unsigned long long count, iter_max;
    double *out;
    unsigned long long *out_split;
    Frame *stack;
    uint32_t split_words, max_len, n_lanes;
};

__global__ void __launch_bounds__(256) k_score_words(DevModel m, WordsArgs a) {
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int c_probe = 0, c_step = 0, c_long = 0;
    Frame *st = a.stack + lane;  // frame of depth L at st[L * n_lanes]
    const size_t fs = a.n_lanes;
    for (unsigned long long t = lane; t < a.count; t += a.n_lanes) {
        const unsigned long long p0 = a.off[t];
        const uint32_t len = (uint32_t)(a.off[t + 1] - p0);
        if (a.out_split)
            for (uint32_t w = 0; w < a.split_words; ++w) a.out_split[t * a.split_words + w] = 0;
        if (len > a.max_len) {
            a.out[t] = __longlong_as_double(0x7FF8000000000000ll);
            ++c_long;
            continue;
        }
        const uint32_t total = len + m.end_text_len;  // len(pwd) + len(end_chr)
        auto ch = [&](uint32_t j) -> uint32_t {
            return j < len ? a.cps[p0 + j] : m.sym_text[m.end_text_off + (j - len)];
        };
        double best = 1022.0;
        unsigned long long done = 0;
        uint32_t depth = 0;
        if (a.iter_max > 0) {  // the entry check of the outermost call (:39)
            st[0] = Frame{0.0, m.start_state, 0, 1};
            depth = 1;
        }
        while (depth > 0) {
            Frame f = st[(depth - 1) * fs];
            const uint32_t rest = total - f.pos;
            const uint32_t top = rest < m.max_word_len ? rest : m.max_word_len;  // no symbol is longer
            unsigned long long h = 0xCBF29CE484222325ull;
            for (uint32_t j = 0; j + 1 < f.idx && j < top; ++j) h = (h ^ ch(f.pos + j)) * 0x100000001B3ull;
            bool advanced = false;
            for (uint32_t idx = f.idx; idx <= top; ++idx) {
                h = (h ^ ch(f.pos + idx - 1)) * 0x100000001B3ull;
                // `left in self.words`: the symbol whose text is text[pos : pos + idx]
                uint32_t sym = MC_NONE;
                {
                    const unsigned long long key = h == ~0ull ? ~0ull - 1 : h;
                    for (unsigned long long q = key & m.word_mask;; q = (q + 1) & m.word_mask) {
                        const Slot sl = m.wordhash[q];
                        if (sl.key == ~0ull) break;
                        if (sl.key == key && sl.pad == idx) {
                            const uint32_t *txt = m.sym_text + m.sym_text_off[sl.val];
                            bool same = true;
                            for (uint32_t j = 0; j < idx; ++j) same = same && txt[j] == ch(f.pos + j);
                            if (same) {
                                sym = sl.val;
                                break;
                            }
                        }
                    }
                }
                // a start symbol is dropped from the components (:50), so a parse through it never
                // reaches target_len: that subtree neither changes the minimum nor the parse count
                if (sym == MC_NONE || sym == m.start_sym) continue;
                const uint32_t e = find_entry(m, f.state, sym, c_probe);
                if (e == MC_NONE) continue;
                const Payload p = m.payload[e];
                const double nacc = f.acc + p.ml2p;
                ++c_step;
                st[(depth - 1) * fs].idx = (uint16_t)(idx + 1);
                if (f.pos + idx == total) {  // a full parse (:51-56)
                    if (best > nacc) {
                        best = nacc;
                        if (a.out_split) {
                            unsigned long long *sp = a.out_split + t * a.split_words;
                            for (uint32_t w = 0; w < a.split_words; ++w) sp[w] = 0;
                            for (uint32_t l = 0; l < depth; ++l) {
                                const Frame g = st[l * fs];
                                const uint32_t end = g.pos + g.idx - 1u;  // idx already points past the split taken
                                if (((end - 1) >> 6) < a.split_words) sp[(end - 1) >> 6] |= 1ull << ((end - 1) & 63);
                            }
                        }
                    }
                    ++done;
                }
                if (done < a.iter_max) {  // the callee's entry check (:39)
                    st[depth * fs] = Frame{nacc, p.next, (uint16_t)(f.pos + idx), 1};
                    ++depth;
                }
                advanced = true;
                break;
            }
            if (!advanced) --depth;
        }
        a.out[t] = best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_probe += __shfl_down(c_probe, o);
        c_step += __shfl_down(c_step, o);
        c_long += __shfl_down(c_long, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_PROBES], (unsigned long long)c_probe);
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_STEPS], (unsigned long long)c_step);
        if (c_long) atomicAdd(&status_slot(m)[MC_CNT_SCORE_TOO_LONG], (unsigned long long)c_long);
    }
}

// ---- the same DFS for short passwords: at most WF_MAX characters with the end symbol's, symbols of at most WF_MAX.
// k_score_words above is a chain of some 600 dependent loads a password (characters, one probe after another, the
// stack in memory); here a lane keeps its password's characters in LDS, looks up ALL lengths of a position at once
// (independent gathers), keeps the frame it works on in registers, stores a frame only when it leaves candidates
// behind (a position with one candidate -- the common case -- is a tail call) and checks a candidate's text beside
// its transition look-up.  Same order of the search, same running sums, same counters of parses: same results.
//   LDS: character j of the lane's password at s_ch[j * 256], symbol of the candidate of length k + 1 at s_sym[k * 256]
//   a stored frame: sum so far, automaton state, lengths not tried yet, split ends of the path (its highest bit = pos - 1)
constexpr uint32_t WF_MAX = 32;
#ifndef WF_WAVES
#define WF_WAVES 4
#endif
struct alignas(8) WFrame {
    double acc;
    uint32_t state, rem, path, pad;
};
static_assert(sizeof(WFrame) == 24, "frame layout");

constexpr unsigned long long FNV_BASIS = 0xCBF29CE484222325ull, FNV_PRIME = 0x100000001B3ull;

// text[pos : pos + idx] in the symbol table, probe by probe with its text compared (a hit of the batched look-up whose
// text differs: two texts of one length with one 64-bit hash)
__device__ __forceinline__ uint32_t words_lookup_exact(const DevModel &m, const uint32_t *s_ch, uint32_t pos, uint32_t idx) {
    unsigned long long h = FNV_BASIS;
    for (uint32_t j = 0; j < idx; ++j) h = (h ^ s_ch[(pos + j) * 256u]) * FNV_PRIME;
    const unsigned long long key = h == ~0ull ? ~0ull - 1 : h;
    for (unsigned long long q = key & m.word_mask;; q = (q + 1) & m.word_mask) {
        const Slot sl = m.wordhash[q];
        if (sl.key == ~0ull) return MC_NONE;
        if (sl.key == key && sl.pad == idx) {
            const uint32_t *txt = m.sym_text + m.sym_text_off[sl.val];
            bool same = true;
            for (uint32_t j = 0; j < idx; ++j) same = same && txt[j] == s_ch[(pos + j) * 256u];
            if (same) return sl.val;
        }
    }
}

template <uint32_t CH>  // lengths looked up together: the model's longest symbol when that is below 8
__global__ void __launch_bounds__(256, WF_WAVES) k_score_words_fast(DevModel m, WordsArgs a, uint32_t rows) {
    extern __shared__ uint32_t wf_lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t *const s_ch = wf_lds + tid;
    uint32_t *const s_sym = wf_lds + rows * 256u + tid;
    const uint32_t lane = blockIdx.x * 256u + tid;
    WFrame *const st = (WFrame *)a.stack + lane;  // stored frame number L at st[L * n_lanes]
    const size_t fs = a.n_lanes;
    const uint32_t W = m.max_word_len;
    unsigned int c_probe = 0, c_step = 0, c_long = 0;

    unsigned long long t_next = lane, t = 0, done = 0;
    // (the offsets of a lane's next password are fetched while it works on the one before)
    unsigned long long p0_next = 0, p1_next = 0;
    if (t_next < a.count) {
        p0_next = a.off[t_next];
        p1_next = a.off[t_next + 1];
    }
    bool have = false, fresh = false;
    uint32_t total = 0, depth = 0, best_path = 0, state = 0, pos = 0, path = 0, mask = 0, filter = 0;
    double best = 0.0, acc = 0.0;
    for (;;) {
        if (!have) {  // the next password of this lane
            if (t_next >= a.count) break;
            t = t_next;
            t_next += a.n_lanes;
            const unsigned long long p0 = p0_next;
            const uint32_t len = (uint32_t)(p1_next - p0);
            if (t_next < a.count) {
                p0_next = a.off[t_next];
                p1_next = a.off[t_next + 1];
            }
            if (a.out_split)
                for (uint32_t w = 0; w < a.split_words; ++w) a.out_split[t * a.split_words + w] = 0;
            if (len > a.max_len) {
                a.out[t] = __longlong_as_double(0x7FF8000000000000ll);
                ++c_long;
                continue;
            }
            total = len + m.end_text_len;  // len(pwd) + len(end_chr)
            for (uint32_t j0 = 0; j0 < total; j0 += 8) {
                uint32_t c[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t j = j0 + u;
                    c[u] = j < len ? a.cps[p0 + j] : (j < total ? m.sym_text[m.end_text_off + (j - len)] : 0u);
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u)
                    if (j0 + u < total) s_ch[(j0 + u) * 256u] = c[u];
            }
            best = 1022.0;
            best_path = 0;
            done = 0;
            depth = 0;
            acc = 0.0;
            state = m.start_state;
            pos = 0;
            path = 0;
            filter = ~0u;
            mask = 0;
            have = fresh = true;
            if (a.iter_max == 0) {  // the entry check of the outermost call (:39)
                a.out[t] = best;
                have = false;
                continue;
            }
        }
        if (fresh) {
            // every length of a symbol that can start at pos, looked up together: FNV-1a runs over the characters,
            // both slots of every length in the two-choice table are independent gathers -- one round trip
            const uint32_t rest = total - pos, top = rest < W ? rest : W;
            unsigned long long h = FNV_BASIS;
            uint32_t found = 0;
            for (uint32_t base = 0; base < top; base += CH) {
                unsigned long long key[CH];
                uint32_t need = 0;
#pragma unroll
                for (uint32_t u = 0; u < CH; ++u) {
                    key[u] = 0;
                    if (base + u < top) {
                        h = (h ^ s_ch[(pos + base + u) * 256u]) * FNV_PRIME;
                        key[u] = h == ~0ull ? ~0ull - 1 : h;
                        need |= ((filter >> (base + u)) & 1u) << u;
                    }
                }
                unsigned long long s1[CH], s2[CH];
#pragma unroll
                for (uint32_t u = 0; u < CH; ++u)
                    if ((need >> u) & 1u) {
                        s1[u] = m.wordck[MC_CK_POS1(key[u]) & m.wordck_mask];
                        s2[u] = m.wordck[MC_CK_POS2(key[u]) & m.wordck_mask];
                    }
#pragma unroll
                for (uint32_t u = 0; u < CH; ++u)
                    if ((need >> u) & 1u) {
                        const uint32_t tag = MC_CK_TAG(key[u], base + u + 1u);
                        const bool in1 = (uint32_t)(s1[u] >> 32) == tag && (uint32_t)s1[u] != MC_NONE;
                        const bool in2 = (uint32_t)(s2[u] >> 32) == tag && (uint32_t)s2[u] != MC_NONE;
                        if (in1 || in2) {
                            found |= 1u << (base + u);
                            s_sym[(base + u) * 256u] = (uint32_t)(in1 ? s1[u] : s2[u]);
                        }
                    }
            }
            mask = found & filter;
            fresh = false;
        }
        if (mask == 0) {  // this position is done: back to the last frame that left candidates behind
            if (depth == 0) {
                a.out[t] = best;
                if (a.out_split) a.out_split[t * a.split_words] = best_path;
                have = false;
                continue;
            }
            --depth;
            const WFrame f = st[depth * fs];
            acc = f.acc;
            state = f.state;
            filter = f.rem;
            path = f.path;
            pos = path ? 32u - (uint32_t)__clz(path) : 0u;
            fresh = true;  // (its candidates' symbols are looked up again: one round of gathers)
            continue;
        }
        const uint32_t k = (uint32_t)__ffs(mask) - 1u, idx = k + 1u;
        mask &= mask - 1u;
        uint32_t sym = s_sym[k * 256u];
        // `left in self.words`: the table's hit is by hash and length; its text is compared beside the transition
        // look-up, which a hit by another text (never seen) would repeat
        const uint32_t toff = m.sym_text_off[sym];
        uint32_t e = find_entry_ahead(m, state, sym, c_probe);
        Payload p = m.payload[e == MC_NONE ? 0u : e];
        uint32_t diff = 0;
        for (uint32_t j0 = 0; j0 < idx; j0 += 8) {
            uint32_t tx[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) tx[u] = j0 + u < idx ? m.sym_text[toff + j0 + u] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
                if (j0 + u < idx) diff |= tx[u] ^ s_ch[(pos + j0 + u) * 256u];
        }
        if (diff) {
            sym = words_lookup_exact(m, s_ch, pos, idx);
            if (sym == MC_NONE) continue;
            e = find_entry_ahead(m, state, sym, c_probe);
            p = m.payload[e == MC_NONE ? 0u : e];
        }
        // a start symbol is dropped from the components (:50), so a parse through it never reaches target_len
        if (sym == m.start_sym || e == MC_NONE) continue;
        const double nacc = acc + p.ml2p;
        ++c_step;
        const uint32_t end = pos + idx;
        if (end == total) {  // a full parse (:51-56); the call it would enter has nothing left to split
            if (best > nacc) {
                best = nacc;
                best_path = path | (1u << (end - 1u));
            }
            ++done;
            continue;
        }
        if (done < a.iter_max) {  // the callee's entry check (:39)
            if (mask) {
                st[depth * fs] = WFrame{acc, state, mask, path, 0u};
                ++depth;
            }
            acc = nacc;
            state = p.next;
            pos = end;
            path |= 1u << (end - 1u);
            filter = ~0u;
            fresh = true;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_probe += __shfl_down(c_probe, o);
        c_step += __shfl_down(c_step, o);
        c_long += __shfl_down(c_long, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_PROBES], (unsigned long long)c_probe);
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_STEPS], (unsigned long long)c_step);
        if (c_long) atomicAdd(&status_slot(m)[MC_CNT_SCORE_TOO_LONG], (unsigned long long)c_long);
    }
}

// ---- BPE-PCFG (bpe_simulator.py:104-157) -----------------------------------------------------------
// One draw of the bucketed table for the row whose buckets start at goff (see k_sample).
template <bool INJECT>
__device__ __forceinline__ DrawRec draw_one(const DevModel &m, uint32_t goff, uint32_t gbits, unsigned long long m53,
                                            double rin, unsigned int &c_probes, unsigned int &c_indirect) {
    uint32_t k;
    if (INJECT) {
        const double kk = rin * (double)(1u << gbits);
        const uint32_t kmax = (1u << gbits) - 1u;
        k = kk >= (double)kmax ? kmax : (kk > 0.0 ? (uint32_t)kk : 0u);
    } else {
        k = (uint32_t)(m53 >> (53u - gbits));
    }
    DrawRec q = ((const DrawRec *)m.draw)[goff + k];
    const uint32_t type = MC_DRAW_TYPE(q.w3);
    if (type != MC_DRAW_DIRECT) {
        const double r = INJECT ? rin : (double)m53 * 0x1.0p-53;
        uint32_t idx = q.w2;
        if (type == MC_DRAW_ONE) {
            idx += (r >= rec_f64(q)) ? 1u : 0u;
            ++c_probes;
        } else {
            uint32_t lo = 0, hi = q.w0;
            const double *T = m.rcum + q.w2;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (T[mid] > r) hi = mid; else lo = mid + 1;
                ++c_probes;
            }
            idx += lo;
        }
        q = ((const DrawRec *)m.spay)[idx];
        ++c_indirect;
    }
    return q;
}

// BpePcfgSim.sample1 (bpe_simulator.py:105-116): draw a structure from the grammar row, then one
// terminal per segment from the row of (tag, t_len); ml2p = the precomputed -log2 p added in draw
// order; no rejection.  One lane per sample; draw k of sample i uses uniform (seed, i, k).
template <bool INJECT>
__global__ void __launch_bounds__(256) k_sample_bpe(DevModel m, SampleArgs a) {
    unsigned int c_draws = 0, c_probes = 0, c_fail = 0, c_indirect = 0;
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long sid = a.first_id + i;
        uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, draw = 0;
        auto next_uniform = [&](unsigned long long &m53, double &r) {
            if (INJECT) {
                r = a.uniforms[i * a.max_draws + draw];
                m53 = 0;
            } else {
                if ((draw & 1u) == 0)
                    philox4x32_10((uint32_t)sid, (uint32_t)(sid >> 32), draw >> 1, 0u, k0, k1, o0, o1, o2, o3);
                const uint32_t wa = (draw & 1u) ? o2 : o0, wb = (draw & 1u) ? o3 : o1;
                m53 = ((unsigned long long)(wa >> 5) << 26) | (unsigned long long)(wb >> 6);
                r = 0.0;
            }
            ++draw;
        };
        unsigned long long m53;
        double r;
        next_uniform(m53, r);
        DrawRec q = draw_one<INJECT>(m, m.start_draw, m.start_gbits, m53, r, c_probes, c_indirect);
        // prob = .0; prob += p (bpe_simulator.py:107-109): the sum STARTS at +0.0 -- a structure of probability 1 is stored
        // as -log2(1) = -0.0, and -0.0 alone (or plus the -0.0 of terminals of probability 1) would stay -0.0, which is
        // not the reference's bits and sorts above every other key (scripts/model_fuzz.py found it: one structure, one terminal)
        double acc = 0.0 + rec_f64(q);
        const uint32_t st = q.w2;  // entry of row 0 = structure index
        const uint32_t s0 = m.seg_off[st], s1 = m.seg_off[st + 1];
        bool ok = true;
        for (uint32_t k = s0; k < s1; ++k) {
            const uint32_t row = m.seg_row[k];
            if (row == MC_NONE || draw >= a.max_draws) {  // self.__terminals[tag_len] raises KeyError (:112)
                ok = false;
                break;
            }
            const StateRec rec = m.state[row];
            next_uniform(m53, r);
            q = draw_one<INJECT>(m, rec.guide_off, rec.gbits, m53, r, c_probes, c_indirect);
            acc += rec_f64(q);
            if (a.out_syms && k - s0 < a.sym_stride) a.out_syms[i * a.sym_stride + (k - s0)] = q.w2;
        }
        c_draws += draw;
        if (!ok) ++c_fail;
        a.out_ml2p[i] = ok ? acc : nan;
        if (a.out_len) a.out_len[i] = s1 - s0;
        if (a.out_draws) a.out_draws[i] = draw;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_draws += __shfl_down(c_draws, o);
        c_probes += __shfl_down(c_probes, o);
        c_fail += __shfl_down(c_fail, o);
        c_indirect += __shfl_down(c_indirect, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_DRAWS], (unsigned long long)c_draws);
        atomicAdd(&status_slot(m)[MC_CNT_PROBES], (unsigned long long)c_probes);
        atomicAdd(&status_slot(m)[MC_CNT_INDIRECT], (unsigned long long)c_indirect);
        if (c_fail) atomicAdd(&status_slot(m)[MC_CNT_FAILED], (unsigned long long)c_fail);
    }
}

// BpePcfgSim.calc_ml2p (bpe_simulator.py:118-150) for one password, one load after another (the walk as written:
// k_score_bpe, and k_score_bpe_fast for passwords longer than its LDS rows)
__device__ double bpe_score_one(const DevModel &m, const uint32_t *__restrict__ cps, const uint8_t *__restrict__ cls,
                                unsigned long long p0, uint32_t len, unsigned int &c_probe, unsigned int &c_step) {
    // luds (:17-40): runs of equal class; hashed as the label index hashes its units (mc_ctx_hash).
    // luds of "" is ((" ", 0),): no structure has that label, and no structure has total length 0.
    uint32_t n_units = 0;
    for (uint32_t j = 0; j < len; ++j)
        if (j == 0 || cls[p0 + j] != cls[p0 + j - 1]) ++n_units;
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)n_units << 32);
    for (uint32_t j = 0; j < len;) {
        uint32_t e = j + 1;
        while (e < len && cls[p0 + e] == cls[p0 + j]) ++e;
        const uint32_t unit = ((uint32_t)cls[p0 + j] << 24) | (e - j);
        h = mix64(h ^ ((unsigned long long)unit + 0x632BE59BD9B4E019ull));
        j = e;
    }
    if (h == ~0ull) h = ~0ull - 1;
    const uint32_t *cand = nullptr;
    uint32_t n_cand = 0;
    if (len > 0 && m.lab_mask) {
        for (unsigned long long q = h & m.lab_mask;; q = (q + 1) & m.lab_mask) {
            const Slot sl = m.labhash[q];
            if (sl.key == ~0ull) break;
            if (sl.key == h && sl.pad == n_units) {
                const uint32_t *u = m.lab_units + m.lab_off[sl.val];
                bool same = true;
                uint32_t ui = 0;
                for (uint32_t j = 0; j < len && same;) {
                    uint32_t e = j + 1;
                    while (e < len && cls[p0 + e] == cls[p0 + j]) ++e;
                    same = u[ui++] == (((uint32_t)cls[p0 + j] << 24) | (e - j));
                    j = e;
                }
                if (same) {
                    cand = m.lab_cand + m.lab_cand_off[sl.val];
                    n_cand = m.lab_cand_off[sl.val + 1] - m.lab_cand_off[sl.val];
                    break;
                }
            }
        }
    }
    if (n_cand == 0)  // :122-125: fall back to the mixed-tag structures of the same total length
        for (uint32_t k = 0; k < m.n_np; ++k)
            if (m.np_len[k] == len) {
                cand = m.np_cand + m.np_off[k];
                n_cand = m.np_off[k + 1] - m.np_off[k];
                break;
            }
    double best = 1022.0;
    for (uint32_t c = 0; c < n_cand; ++c) {
        const uint32_t st = cand[c];
        double p = m.payload[st].ml2p;  // grammars[candidate]: row 0 starts at entry 0
        uint32_t start = 0;
        bool ok = true;
        for (uint32_t k = m.seg_off[st]; k < m.seg_off[st + 1]; ++k) {
            const uint32_t row = m.seg_row[k], tl = m.seg_len[k];
            // replacement = pwd[start:start + t_len]  (slicing clamps at the end of the string)
            const uint32_t b = start < len ? start : len, e = start + tl < len ? start + tl : len;
            start += tl;
            uint32_t sym = MC_NONE;
            if (row != MC_NONE && e > b) {
                unsigned long long hh = 0xCBF29CE484222325ull;
                for (uint32_t j = b; j < e; ++j) hh = (hh ^ cps[p0 + j]) * 0x100000001B3ull;
                if (hh == ~0ull) hh = ~0ull - 1;
                for (unsigned long long q = hh & m.word_mask;; q = (q + 1) & m.word_mask) {
                    const Slot sl = m.wordhash[q];
                    if (sl.key == ~0ull) break;
                    if (sl.key == hh && sl.pad == e - b) {
                        const uint32_t *txt = m.sym_text + m.sym_text_off[sl.val];
                        bool same = true;
                        for (uint32_t j = 0; j < e - b; ++j) same = same && txt[j] == cps[p0 + b + j];
                        if (same) {
                            sym = sl.val;
                            break;
                        }
                    }
                }
            }
            uint32_t ent = MC_NONE;
            if (sym != MC_NONE) {
                const unsigned long long key = ((unsigned long long)row << 32) | sym;
                for (unsigned long long q = mix64(key + 0x9E3779B97F4A7C15ull) & m.trans_mask;;
                     q = (q + 1) & m.trans_mask) {
                    const Slot sl = m.trans[q];
                    ++c_probe;
                    if (sl.key == key) {
                        ent = sl.val;
                        break;
                    }
                    if (sl.key == ~0ull) break;
                }
            }
            if (ent == MC_NONE) {  // replacement not in terminal (:139-141)
                ok = false;
                break;
            }
            p += m.payload[ent].ml2p;
            ++c_step;
        }
        if (ok && p < 1022.0 && p < best) best = p;
    }
    return best;
}

__global__ void __launch_bounds__(256) k_score_bpe(DevModel m, const uint32_t *__restrict__ cps,
                                                   const uint8_t *__restrict__ cls,
                                                   const unsigned long long *__restrict__ off, unsigned long long count,
                                                   double *__restrict__ out) {
    unsigned int c_probe = 0, c_step = 0;
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < count;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long p0 = off[t];
        out[t] = bpe_score_one(m, cps, cls, p0, (uint32_t)(off[t + 1] - p0), c_probe, c_step);
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_probe += __shfl_down(c_probe, o);
        c_step += __shfl_down(c_step, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_PROBES], (unsigned long long)c_probe);
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_STEPS], (unsigned long long)c_step);
    }
}

// ---- the same for passwords of at most BF_MAX characters, a load ROUND after another: characters and classes in LDS
// (class << 24 | code point at s_cc[j * 256]), the class runs from a bit mask of their first characters, the label's
// units compared in one round, and per candidate structure the segments four at a time: their rows and lengths, both
// slots of every terminal in the two-choice symbol table, the transitions' first probes beside the texts' offsets, the
// payloads beside the texts -- five rounds where the walk above makes some fifty dependent loads.  Same sums in the
// same order.  A terminal whose text differs from the hit's (two texts, one hash) is looked up again probe by probe.
constexpr uint32_t BF_MAX = 32, BF_SEG = 4;

__device__ __forceinline__ uint32_t bf_unit(const uint32_t *s_cc, uint32_t &rem, uint32_t len) {  // the next class run
    const uint32_t s0 = (uint32_t)__ffs(rem) - 1u;
    rem &= rem - 1u;
    const uint32_t e0 = rem ? (uint32_t)__ffs(rem) - 1u : len;
    return (s_cc[s0 * 256u] & 0xFF000000u) | (e0 - s0);
}

__device__ __forceinline__ uint32_t bf_find_trans(const DevModel &m, uint32_t row, uint32_t sym, unsigned int &c_probe) {
    const unsigned long long key = ((unsigned long long)row << 32) | sym;
    for (unsigned long long q = mix64(key + 0x9E3779B97F4A7C15ull) & m.trans_mask;; q = (q + 1) & m.trans_mask) {
        const Slot sl = m.trans[q];
        ++c_probe;
        if (sl.key == key) return sl.val;
        if (sl.key == ~0ull) return MC_NONE;
    }
}

__global__ void __launch_bounds__(256) k_score_bpe_fast(DevModel m, const uint32_t *__restrict__ cps,
                                                        const uint8_t *__restrict__ cls,
                                                        const unsigned long long *__restrict__ off,
                                                        unsigned long long count, double *__restrict__ out) {
    __shared__ uint32_t bf_lds[BF_MAX * 256];
    uint32_t *const s_cc = bf_lds + threadIdx.x;
    unsigned int c_probe = 0, c_step = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256u;
    unsigned long long t = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    unsigned long long p0n = 0, p1n = 0;  // (a lane's next offsets are requested a password ahead)
    if (t < count) {
        p0n = off[t];
        p1n = off[t + 1];
    }
    for (; t < count; t += stride) {
        const unsigned long long p0 = p0n;
        const uint32_t len = (uint32_t)(p1n - p0);
        if (t + stride < count) {
            p0n = off[t + stride];
            p1n = off[t + stride + 1];
        }
        if (len == 0 || len > BF_MAX) {
            out[t] = bpe_score_one(m, cps, cls, p0, len, c_probe, c_step);
            continue;
        }
        uint32_t bm = 0, prev = ~0u;  // bit j: character j starts a class run (luds, :17-40)
        for (uint32_t j0 = 0; j0 < len; j0 += 8) {
            uint32_t c[8], k[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                c[u] = j0 + u < len ? cps[p0 + j0 + u] : 0u;
                k[u] = j0 + u < len ? (uint32_t)cls[p0 + j0 + u] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
                if (j0 + u < len) {
                    s_cc[(j0 + u) * 256u] = (k[u] << 24) | c[u];
                    bm |= (k[u] != prev ? 1u : 0u) << (j0 + u);
                    prev = k[u];
                }
        }
        const uint32_t n_units = (uint32_t)__popc(bm);
        unsigned long long h = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)n_units << 32);
        for (uint32_t rem = bm; rem;) h = mix64(h ^ ((unsigned long long)bf_unit(s_cc, rem, len) + 0x632BE59BD9B4E019ull));
        if (h == ~0ull) h = ~0ull - 1;
        const uint32_t *cand = nullptr;
        uint32_t n_cand = 0;
        if (m.lab_mask) {
            for (unsigned long long q = h & m.lab_mask;; q = (q + 1) & m.lab_mask) {
                const Slot sl = m.labhash[q];
                if (sl.key == ~0ull) break;
                if (sl.key == h && sl.pad == n_units) {
                    const uint32_t lo = m.lab_off[sl.val], c0 = m.lab_cand_off[sl.val], c1 = m.lab_cand_off[sl.val + 1];
                    const uint32_t *un = m.lab_units + lo;
                    uint32_t diff = 0, rem = bm;
                    for (uint32_t k0 = 0; k0 < n_units; k0 += 8) {
                        uint32_t uu[8];
#pragma unroll
                        for (uint32_t i = 0; i < 8; ++i) uu[i] = k0 + i < n_units ? un[k0 + i] : 0u;
#pragma unroll
                        for (uint32_t i = 0; i < 8; ++i)
                            if (k0 + i < n_units) diff |= uu[i] ^ bf_unit(s_cc, rem, len);
                    }
                    if (!diff) {
                        cand = m.lab_cand + c0;
                        n_cand = c1 - c0;
                        break;
                    }
                }
            }
        }
        if (n_cand == 0)  // :122-125: fall back to the mixed-tag structures of the same total length
            for (uint32_t k = 0; k < m.n_np; ++k)
                if (m.np_len[k] == len) {
                    cand = m.np_cand + m.np_off[k];
                    n_cand = m.np_off[k + 1] - m.np_off[k];
                    break;
                }
        double best = 1022.0;
        for (uint32_t c = 0; c < n_cand; ++c) {
            const uint32_t st = cand[c];
            double p = m.payload[st].ml2p;  // grammars[candidate]: row 0 starts at entry 0
            const uint32_t k0 = m.seg_off[st], k1 = m.seg_off[st + 1];
            uint32_t start = 0;
            bool ok = true;
            for (uint32_t kb = k0; kb < k1 && ok; kb += BF_SEG) {
                uint32_t row[BF_SEG], tl[BF_SEG], bb[BF_SEG], ee[BF_SEG], sym[BF_SEG], ent[BF_SEG], toff[BF_SEG], diff[BF_SEG];
                unsigned long long key[BF_SEG], s1[BF_SEG], s2[BF_SEG];
                Slot ts[BF_SEG];
                double pl[BF_SEG];
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i) {
                    row[i] = kb + i < k1 ? m.seg_row[kb + i] : MC_NONE;
                    tl[i] = kb + i < k1 ? m.seg_len[kb + i] : 0u;
                }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i) {
                    // replacement = pwd[start:start + t_len]  (slicing clamps at the end of the string)
                    bb[i] = start < len ? start : len;
                    ee[i] = start + tl[i] < len ? start + tl[i] : len;
                    start += tl[i];
                    sym[i] = ent[i] = MC_NONE;
                    diff[i] = 0;
                    key[i] = 0;
                    if (row[i] != MC_NONE && ee[i] > bb[i]) {
                        unsigned long long hh = FNV_BASIS;
                        for (uint32_t j = bb[i]; j < ee[i]; ++j) hh = (hh ^ (s_cc[j * 256u] & 0xFFFFFFu)) * FNV_PRIME;
                        key[i] = hh == ~0ull ? ~0ull - 1 : hh;
                        s1[i] = m.wordck[MC_CK_POS1(key[i]) & m.wordck_mask];
                        s2[i] = m.wordck[MC_CK_POS2(key[i]) & m.wordck_mask];
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i)
                    if (row[i] != MC_NONE && ee[i] > bb[i]) {
                        const uint32_t tag = MC_CK_TAG(key[i], ee[i] - bb[i]);
                        if ((uint32_t)(s1[i] >> 32) == tag && (uint32_t)s1[i] != MC_NONE) sym[i] = (uint32_t)s1[i];
                        else if ((uint32_t)(s2[i] >> 32) == tag && (uint32_t)s2[i] != MC_NONE) sym[i] = (uint32_t)s2[i];
                    }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i)
                    if (sym[i] != MC_NONE) {
                        const unsigned long long tk = ((unsigned long long)row[i] << 32) | sym[i];
                        ts[i] = m.trans[mix64(tk + 0x9E3779B97F4A7C15ull) & m.trans_mask];
                        toff[i] = m.sym_text_off[sym[i]];
                    }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i)
                    if (sym[i] != MC_NONE) {
                        const unsigned long long tk = ((unsigned long long)row[i] << 32) | sym[i];
                        ++c_probe;
                        if (ts[i].key == tk) {
                            ent[i] = ts[i].val;
                        } else if (ts[i].key != ~0ull) {  // (another key in the first slot: the walk goes on)
                            for (unsigned long long q = (mix64(tk + 0x9E3779B97F4A7C15ull) + 1) & m.trans_mask;;
                                 q = (q + 1) & m.trans_mask) {
                                const Slot sl = m.trans[q];
                                ++c_probe;
                                if (sl.key == tk) {
                                    ent[i] = sl.val;
                                    break;
                                }
                                if (sl.key == ~0ull) break;
                            }
                        }
                    }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i) {
                    pl[i] = m.payload[ent[i] == MC_NONE ? 0u : ent[i]].ml2p;
                    if (sym[i] != MC_NONE) {
                        const uint32_t n = ee[i] - bb[i];
                        for (uint32_t j0 = 0; j0 < n; j0 += 8) {
                            uint32_t tx[8];
#pragma unroll
                            for (uint32_t u = 0; u < 8; ++u) tx[u] = j0 + u < n ? m.sym_text[toff[i] + j0 + u] : 0u;
#pragma unroll
                            for (uint32_t u = 0; u < 8; ++u)
                                if (j0 + u < n) diff[i] |= tx[u] ^ (s_cc[(bb[i] + j0 + u) * 256u] & 0xFFFFFFu);
                        }
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < BF_SEG; ++i) {
                    if (!ok || kb + i >= k1) continue;
                    if (diff[i]) {  // the hit's text is another one's: the terminal is looked up again, probe by probe
                        ent[i] = MC_NONE;
                        const uint32_t n = ee[i] - bb[i];
                        for (unsigned long long q = key[i] & m.word_mask;; q = (q + 1) & m.word_mask) {
                            const Slot sl = m.wordhash[q];
                            if (sl.key == ~0ull) break;
                            if (sl.key == key[i] && sl.pad == n) {
                                const uint32_t *txt = m.sym_text + m.sym_text_off[sl.val];
                                bool same = true;
                                for (uint32_t j = 0; j < n; ++j) same = same && txt[j] == (s_cc[(bb[i] + j) * 256u] & 0xFFFFFFu);
                                if (same) {
                                    ent[i] = bf_find_trans(m, row[i], sl.val, c_probe);
                                    break;
                                }
                            }
                        }
                        pl[i] = m.payload[ent[i] == MC_NONE ? 0u : ent[i]].ml2p;
                    }
                    if (ent[i] == MC_NONE) {  // replacement not in terminal (:139-141)
                        ok = false;
                    } else {
                        p += pl[i];
                        ++c_step;
                    }
                }
            }
            if (ok && p < 1022.0 && p < best) best = p;
        }
        out[t] = best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_probe += __shfl_down(c_probe, o);
        c_step += __shfl_down(c_step, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_PROBES], (unsigned long long)c_probe);
        atomicAdd(&status_slot(m)[MC_CNT_SCORE_STEPS], (unsigned long long)c_step);
    }
}

// ---- GPU-side `sampled_pwds` (MonteCarloParent.py:40-45): string -> (first sample, occurrences) ----
__device__ __forceinline__ unsigned long long row_hash(const uint32_t *row, uint32_t n) {
    unsigned long long h = 0xCBF29CE484222325ull ^ n;
    for (uint32_t j = 0; j < n; ++j) h = (h ^ row[j]) * 0x100000001B3ull;
    h = mix64(h);
    return h == ~0ull ? ~0ull - 1 : h;
}

__global__ void __launch_bounds__(256) k_dedup_insert(const uint32_t *__restrict__ syms, const uint32_t *__restrict__ len,
                                                      uint32_t stride, unsigned long long count,
                                                      unsigned long long mask, unsigned long long *tab_key,
                                                      unsigned long long *tab_first, uint32_t *tab_cnt) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t n = len[i];
        if (n > stride) continue;  // truncated row: the caller replays this sample
        const unsigned long long h = row_hash(syms + i * stride, n);
        for (unsigned long long q = h & mask;; q = (q + 1) & mask) {
            const unsigned long long old = atomicCAS(&tab_key[q], ~0ull, h);
            if (old == ~0ull || old == h) {
                atomicMin(&tab_first[q], i);
                atomicAdd(&tab_cnt[q], 1u);
                break;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_dedup_verify(const uint32_t *__restrict__ syms, const uint32_t *__restrict__ len,
                                                      uint32_t stride, unsigned long long count,
                                                      unsigned long long mask,
                                                      const unsigned long long *__restrict__ tab_key,
                                                      const unsigned long long *__restrict__ tab_first,
                                                      uint32_t *mismatch) {
    unsigned int bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t n = len[i];
        if (n > stride) continue;
        const uint32_t *row = syms + i * stride;
        const unsigned long long h = row_hash(row, n);
        unsigned long long q = h & mask;
        while (tab_key[q] != h) q = (q + 1) & mask;  // inserted by k_dedup_insert
        const unsigned long long rep = tab_first[q];
        if (rep == i) continue;
        const uint32_t *other = syms + rep * stride;
        bool same = len[rep] == n;
        for (uint32_t j = 0; j < n && same; ++j) same = row[j] == other[j];
        if (!same) ++bad;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatch, bad);
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

struct mc_model {
    int device;
    int n_cu;
    unsigned long long n_draw;  // records in the draw table
    bool has_text;
    void *arena;
    size_t arena_bytes;
    DevModel dev;
    bool multichar, len_in_rec;
    // mc_sample_tune: launch geometry of k_sample (0 / negative: the library's defaults)
    int tune_wg_per_cu = 0, tune_hot_kb = -1;
    float tune_skew = -1.0f;
};

extern "C" int mc_model_upload(const mc_table *t, int device, mc_model **out) {
    if (!t || !out) {
        mc_set_error("mc_model_upload: null argument");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    mc_model *m = new (std::nothrow) mc_model();
    if (!m) return MC_ERR_NOMEM;
    m->device = device;
    m->multichar = t->multichar;
    m->len_in_rec = t->len_in_rec;
    m->has_text = !t->wordhash.empty();
    m->n_draw = t->draw.size();
    m->arena_bytes = mc_table_device_bytes(t);
    hipError_t e = hipMalloc(&m->arena, m->arena_bytes);
    if (e != hipSuccess) {
        mc_set_error("hipMalloc(%zu) failed: %s", m->arena_bytes, hipGetErrorString(e));
        delete m;
        (void)hipSetDevice(prev);
        return MC_ERR_NOMEM;
    }
    HIP_TRY(hipDeviceGetAttribute(&m->n_cu, hipDeviceAttributeMultiprocessorCount, device));
    char *p = (char *)m->arena;
    auto put = [&](const void *src, size_t bytes) -> void * {
        void *dst = p;
        if (bytes) (void)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        p += align256(bytes);
        return dst;
    };
    DevModel &d = m->dev;
    d.state = (const StateRec *)put(t->state.data(), t->state.size() * sizeof(mc_state_rec));
    d.draw = (const mc_draw_rec *)put(t->draw.data(), t->draw.size() * sizeof(mc_draw_rec));
    d.spay = (const mc_draw_rec *)put(t->spay.data(), t->spay.size() * sizeof(mc_draw_rec));
    d.rcum = (const double *)put(t->rcum.data(), t->rcum.size() * sizeof(double));
    d.payload = (const Payload *)put(t->payload.data(), t->payload.size() * sizeof(mc_payload));
    d.trans = (const Slot *)put(t->trans.data(), t->trans.size() * sizeof(mc_slot));
    d.ctxhash = (const Slot *)put(t->ctxhash.data(), t->ctxhash.size() * sizeof(mc_slot));
    d.sym_chars = (const uint32_t *)put(t->sym_chars.data(), t->sym_chars.size() * sizeof(uint32_t));
    d.wordhash = (const Slot *)put(t->wordhash.data(), t->wordhash.size() * sizeof(mc_slot));
    d.sym_text_off = (const uint32_t *)put(t->sym_text_off.data(), t->sym_text_off.size() * sizeof(uint32_t));
    d.sym_text = (const uint32_t *)put(t->sym_text.data(), t->sym_text.size() * sizeof(uint32_t));
    d.word_mask = t->wordhash.empty() ? 0 : t->wordhash.size() - 1;
    d.wordck = t->wordck.empty() ? nullptr : (const unsigned long long *)put(t->wordck.data(), t->wordck.size() * sizeof(uint64_t));
    d.wordck_mask = t->wordck.empty() ? 0 : t->wordck.size() - 1;
    d.max_word_len = t->max_word_len;
    d.start_sym = t->start_sym;
    d.end_text_off = t->sym_text_off.empty() ? 0 : t->sym_text_off[t->end_sym];
    d.end_text_len = t->sym_text_off.empty() ? 0 : t->sym_text_off[t->end_sym + 1] - t->sym_text_off[t->end_sym];
    d.dense = t->dense.empty() ? nullptr : (const Payload *)put(t->dense.data(), t->dense.size() * sizeof(mc_payload));
    d.seg_off = (const uint32_t *)put(t->seg_off.data(), t->seg_off.size() * 4);
    d.seg_row = (const uint32_t *)put(t->seg_row.data(), t->seg_row.size() * 4);
    d.seg_len = (const uint32_t *)put(t->seg_len.data(), t->seg_len.size() * 4);
    d.labhash = (const Slot *)put(t->labhash.data(), t->labhash.size() * sizeof(mc_slot));
    d.lab_off = (const uint32_t *)put(t->lab_off.data(), t->lab_off.size() * 4);
    d.lab_units = (const uint32_t *)put(t->lab_units.data(), t->lab_units.size() * 4);
    d.lab_cand_off = (const uint32_t *)put(t->lab_cand_off.data(), t->lab_cand_off.size() * 4);
    d.lab_cand = (const uint32_t *)put(t->lab_cand.data(), t->lab_cand.size() * 4);
    d.np_len = (const uint32_t *)put(t->np_len.data(), t->np_len.size() * 4);
    d.np_off = (const uint32_t *)put(t->np_off.data(), t->np_off.size() * 4);
    d.np_cand = (const uint32_t *)put(t->np_cand.data(), t->np_cand.size() * 4);
    d.lab_mask = t->labhash.empty() ? 0 : t->labhash.size() - 1;
    d.n_np = (uint32_t)t->np_len.size();
    d.kind = t->kind;
    d.status = (unsigned long long *)p;
    HIP_TRY(hipMemset(d.status, 0, (size_t)MC_STATUS_SLOTS * MC_STATUS_STRIDE * sizeof(uint64_t)));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipGetLastError());
    d.trans_mask = t->trans.size() - 1;
    d.ctx_mask = t->ctxhash.size() - 1;
    d.start_state = t->start_state;
    d.root = t->root;
    d.end_sym = t->end_sym;
    d.min_len = t->min_len;
    d.max_len = t->max_len;
    d.flags = t->flags;
    d.n_ctx = t->n_ctx;
    d.n_sym = t->n_sym;
    d.start_draw = t->start_draw;
    d.start_gbits = t->start_gbits;
    HIP_TRY(hipSetDevice(prev));
    *out = m;
    return MC_OK;
}

int mc_model_enum_view(const mc_model *m, mc_enum_view *out) {
    if (!m || !out) {
        mc_set_error("null model");
        return MC_ERR_ARG;
    }
    const DevModel &d = m->dev;
    *out = mc_enum_view{(const mc_state_rec *)d.state, (const mc_payload *)d.payload, d.sym_chars, d.start_state,
                        d.root, d.n_ctx, d.n_sym, d.end_sym, d.kind, m->device};
    return MC_OK;
}

extern "C" void mc_model_destroy(mc_model *m) {
    if (!m) return;
    if (m->arena) (void)hipFree(m->arena);
    delete m;
}

extern "C" int mc_model_check(mc_model *m, void *stream, uint64_t counters[8]) {
    if (!m) {
        mc_set_error("mc_model_check: null model");
        return MC_ERR_ARG;
    }
    static thread_local uint64_t slots[MC_STATUS_SLOTS * MC_STATUS_STRIDE];
    uint64_t host[MC_STATUS_WORDS] = {0};
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(slots, m->dev.status, sizeof slots, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(m->dev.status, 0, sizeof slots, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < MC_STATUS_SLOTS; ++k)
        for (int w = 0; w < MC_STATUS_WORDS; ++w) host[w] += slots[k * MC_STATUS_STRIDE + w];
    if (counters) std::memcpy(counters, host, sizeof host);
    if (host[MC_CNT_SCORE_TOO_LONG]) {
        mc_set_error("%llu password(s) were longer than the max_len given to mc_score_words",
                     (unsigned long long)host[MC_CNT_SCORE_TOO_LONG]);
        return MC_ERR_ARG;
    }
    if (host[MC_CNT_FAILED]) {
        mc_set_error("%llu sample(s) did not reach the end symbol within max_draws (or walked into a missing row); "
                     "the reference would not terminate on this model",
                     (unsigned long long)host[MC_CNT_FAILED]);
        return MC_ERR_DIVERGED;
    }
    return MC_OK;
}

static int launch_sample(mc_model *m, bool inject, const SampleArgs &a_in, hipStream_t s) {
    if (a_in.count == 0) return MC_OK;
    if (m->dev.start_state == MC_NONE) {
        mc_set_error("the start context has no row (the reference exits in pick_expand, ProbLib.py:32-36)");
        return MC_ERR_TABLE;
    }
    SampleArgs a = a_in;
    a.hot_recs = 0;
    const unsigned long long carry_n16 = a_in.carry_n16;
    a.carry_n16 = 0;  // (k_sample_bpe carries nothing: its caller copies)
    a.carry_groups = 0;
    if (m->dev.kind == MC_KIND_BPE) {
        unsigned long long want = (a.count + 255) / 256, cap = (unsigned long long)m->n_cu * 16;
        dim3 g((unsigned)(want < cap ? want : cap)), b(256);
        if (inject) hipLaunchKernelGGL(k_sample_bpe<true>, g, b, 0, s, m->dev, a);
        else hipLaunchKernelGGL(k_sample_bpe<false>, g, b, 0, s, m->dev, a);
        HIP_TRY(hipGetLastError());
        return MC_OK;
    }
    // Workgroups of 256 lanes; 14 per CU, of which 8 are resident at a time: a workgroup owns a
    // contiguous range of sample ids, and the second, partial generation of workgroups takes the slots
    // the first leaves as its waves thin out (the kernel is bound by VALU issue; a wave whose lanes are
    // mostly done costs as much issue as a full one).  Measured at 1e7 samples with the round-2 kernel
    // (scripts/exp_sample_grid.py): 8 per CU 0.497 ms, 12 0.421, 13 0.415, 14 0.410, 15 0.444, 16 0.437,
    // 20 0.426, 32 0.459; 14 is also the best of 12/14/16/20 on the v2, the fixed-order and a smaller
    // backoff model (4-10 % under 16).  The ranges are not equal: they shrink linearly from 1.55 x the mean for the
    // first workgroup to 0.45 x for the last (SampleArgs.skew), so that workgroups end one after the other
    // instead of a generation at a time and the slots they free are taken at once: 0.410 -> 0.382 ms at 1e7
    // samples, 3.46 -> 3.21 ms at 1e8, 5-7 % on the fixed-order and the smaller backoff model, nothing on the
    // v2 model (0.25 .. 0.75 and exponentially shrinking ranges at 14 .. 32 workgroups a CU were tried).
    // 15 KB of LDS per workgroup hold the head of the draw table (with the 4 KB of the waves' key rings 8
    // workgroups fit a CU; it serves about half of the draws of the headline model; 40 KB costs more in
    // occupancy than it saves).  One or two
    // chains per lane perform alike once the table is placed hot-first.  All are tuning knobs
    // (results never depend on them: tests/test_gpu_parity.py).  Each workgroup's share of the
    // sample ids must fit u32.
    static const int env_nc = getenv("MC_SAMPLE_CHAINS") ? atoi(getenv("MC_SAMPLE_CHAINS")) : 1;
    static const int env_wg = getenv("MC_SAMPLE_WG_PER_CU") ? atoi(getenv("MC_SAMPLE_WG_PER_CU")) : 0;
    static const int env_hot = getenv("MC_SAMPLE_HOT_KB") ? atoi(getenv("MC_SAMPLE_HOT_KB")) : 15;
    const int nc = inject ? 1 : env_nc;
    const int hot_kb = m->tune_hot_kb >= 0 ? m->tune_hot_kb : env_hot;
    // Workgroups a CU (8 are resident: 56 registers a lane, 19 KB of LDS a workgroup).  ONE generation for a launch of
    // up to 2^24 samples, two for longer ones, where a second generation takes the slots the first leaves while
    // its waves thin out (scripts/exp_wg_n.py, late round 3, ms at 8 / 12 / 16 a CU -- headline model: 1e6 samples
    // 0.085 / 0.106 / 0.118, 1e7 0.364 / 0.400 / 0.378, 3e7 1.03 / 1.09 / 0.97, 1e8 3.23 / 3.41 / 2.89; v2 6-gram: 1e7
    // 0.262 / 0.300 / 0.274, 1e8 2.37 / 2.25 / 2.25).  With a copy taken along: one generation.
    const int auto_wg = (a_in.carry_n16 || a.count <= (1ull << 24)) ? 8 : 16;
    const int wg_per_cu = m->tune_wg_per_cu > 0 ? m->tune_wg_per_cu : env_wg > 0 ? env_wg : auto_wg;
    {
        unsigned long long recs = (unsigned long long)(hot_kb < 0 ? 0 : hot_kb > 150 ? 150 : hot_kb) * 1024 / 16;
        a.hot_recs = (uint32_t)(recs < m->n_draw ? recs : m->n_draw);
    }
    static const float env_skew = getenv("MC_SAMPLE_SKEW") ? (float)atof(getenv("MC_SAMPLE_SKEW")) : 0.55f;
    const float skew = m->tune_skew >= 0.0f ? m->tune_skew : env_skew;
    a.skew = skew < 0.0f ? 0.0f : skew > 0.95f ? 0.95f : skew;
    const size_t lds = (size_t)a.hot_recs * sizeof(DrawRec);
    unsigned long long want = (a.count + 64ull * nc - 1) / (64ull * nc);
    unsigned long long cap = (unsigned long long)m->n_cu * (wg_per_cu > 0 ? wg_per_cu : 8);
    unsigned long long grid = want < cap ? want : cap;
    while ((a.count + grid - 1) / grid > 0x7FFFFFFFull) grid *= 2;
    if (carry_n16) {
        // one copy workgroup per 2.75 MiB (8 MB: 3 workgroups, 0.2 ms beside 0.38 ms of sampling), at most 32
        static const int env_cg = getenv("MC_CARRY_GROUPS") ? atoi(getenv("MC_CARRY_GROUPS")) : 0;
        const unsigned long long by_size = (carry_n16 * 16 + (11ull << 18) - 1) / (11ull << 18);
        a.carry_n16 = carry_n16;
        a.carry_groups = (uint32_t)(env_cg > 0 ? (env_cg > 256 ? 256 : env_cg) : by_size > 32 ? 32 : by_size);
        grid += a.carry_groups;
    }
    dim3 g((unsigned)grid), b(256);
    const bool extras = a.out_len || a.out_draws || a.out_syms;  // (else the lean kernel: scores only)
    if (inject) {
        if (m->len_in_rec) hipLaunchKernelGGL((k_sample<true, 2, 1>), g, b, lds, s, m->dev, a);
        else if (m->multichar) hipLaunchKernelGGL((k_sample<true, 1, 1>), g, b, lds, s, m->dev, a);
        else hipLaunchKernelGGL((k_sample<true, 0, 1>), g, b, lds, s, m->dev, a);
    } else if (m->len_in_rec) {
        if (nc == 2) hipLaunchKernelGGL((k_sample<false, 2, 2>), g, b, lds, s, m->dev, a);
        else if (extras) hipLaunchKernelGGL((k_sample<false, 2, 1>), g, b, lds, s, m->dev, a);
        else hipLaunchKernelGGL((k_sample<false, 2, 1, false>), g, b, lds, s, m->dev, a);
    } else if (m->multichar) {
        if (nc == 2) hipLaunchKernelGGL((k_sample<false, 1, 2>), g, b, lds, s, m->dev, a);
        else if (extras) hipLaunchKernelGGL((k_sample<false, 1, 1>), g, b, lds, s, m->dev, a);
        else hipLaunchKernelGGL((k_sample<false, 1, 1, false>), g, b, lds, s, m->dev, a);
    } else {
        if (nc == 2) hipLaunchKernelGGL((k_sample<false, 0, 2>), g, b, lds, s, m->dev, a);
        else if (nc == 4) hipLaunchKernelGGL((k_sample<false, 0, 4>), g, b, lds, s, m->dev, a);
        else if (extras) hipLaunchKernelGGL((k_sample<false, 0, 1>), g, b, lds, s, m->dev, a);
        else hipLaunchKernelGGL((k_sample<false, 0, 1, false>), g, b, lds, s, m->dev, a);
    }
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_sample_tune(mc_model *m, int wg_per_cu, float skew, int hot_kb) {
    if (!m || wg_per_cu > 64 || hot_kb > 150 || skew > 0.95f) {
        mc_set_error("mc_sample_tune: null model, or more than 64 workgroups a CU, a skew above 0.95 or more than 150 KB");
        return MC_ERR_ARG;
    }
    m->tune_wg_per_cu = wg_per_cu;
    m->tune_skew = skew;
    m->tune_hot_kb = hot_kb;
    return MC_OK;
}

extern "C" int mc_sample(mc_model *m, uint64_t first_id, uint64_t count, uint64_t seed, uint32_t max_draws,
                         double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
                         uint32_t sym_stride, void *stream) {
    if (!m || (count && !out_ml2p) || max_draws == 0 || (out_syms && sym_stride == 0)) {
        mc_set_error("mc_sample: bad argument");
        return MC_ERR_ARG;
    }
    SampleArgs a{first_id, count, seed, nullptr, out_ml2p, out_len, out_draws, out_syms, max_draws, sym_stride, 0};
    return launch_sample(m, false, a, (hipStream_t)stream);
}

extern "C" int mc_sample_carry(mc_model *m, uint64_t first_id, uint64_t count, uint64_t seed, uint32_t max_draws,
                               double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
                               uint32_t sym_stride, const void *carry_src, void *carry_dst, uint64_t carry_bytes,
                               void *stream) {
    if (!m || (count && !out_ml2p) || max_draws == 0 || (out_syms && sym_stride == 0)) {
        mc_set_error("mc_sample_carry: bad argument");
        return MC_ERR_ARG;
    }
    if (carry_bytes && (!carry_src || !carry_dst || (carry_bytes & 15) || ((uintptr_t)carry_src & 15) ||
                        ((uintptr_t)carry_dst & 15))) {
        mc_set_error("mc_sample_carry: the copy taken along needs two 16-byte aligned addresses and a multiple of 16 "
                     "bytes (got %llu bytes)", (unsigned long long)carry_bytes);
        return MC_ERR_ARG;
    }
    SampleArgs a{first_id, count, seed, nullptr, out_ml2p, out_len, out_draws, out_syms, max_draws, sym_stride, 0};
    const bool in_kernel = carry_bytes && count && m->dev.kind != MC_KIND_BPE && m->dev.start_state != MC_NONE;
    if (in_kernel) {
        a.carry_src = (const uint4 *)carry_src;
        a.carry_dst = (uint4 *)carry_dst;
        a.carry_n16 = carry_bytes / 16;
    }
    const int rc = launch_sample(m, false, a, (hipStream_t)stream);
    if (rc != MC_OK) return rc;
    // (nothing to sample beside, or a BPE model, whose sampler has no such workgroups: the runtime's copy)
    if (carry_bytes && !in_kernel)
        HIP_TRY(hipMemcpyAsync(carry_dst, carry_src, carry_bytes, hipMemcpyDefault, (hipStream_t)stream));
    return MC_OK;
}

extern "C" int mc_sample_injected(mc_model *m, uint64_t count, const double *uniforms, uint32_t draws_per_sample,
                                  double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
                                  uint32_t sym_stride, void *stream) {
    if (!m || (count && (!out_ml2p || !uniforms)) || draws_per_sample == 0 || (out_syms && sym_stride == 0)) {
        mc_set_error("mc_sample_injected: bad argument");
        return MC_ERR_ARG;
    }
    SampleArgs a{0, count, 0, uniforms, out_ml2p, out_len, out_draws, out_syms, draws_per_sample, sym_stride, 0};
    return launch_sample(m, true, a, (hipStream_t)stream);
}

extern "C" int mc_score(mc_model *m, const uint32_t *syms, const uint64_t *off, uint64_t count, double *out_ml2p,
                        void *stream) {
    if (!m || (count && (!off || !out_ml2p))) {
        mc_set_error("mc_score: bad argument");
        return MC_ERR_ARG;
    }
    if (count == 0) return MC_OK;
    if (m->multichar || m->dev.kind == MC_KIND_BPE || m->dev.kind == MC_KIND_V2) {
        mc_set_error("mc_score walks single-character n-gram / backoff alphabets only (sub-word alphabets: "
                     "mc_score_words; BPE-PCFG: mc_score_bpe)");
        return MC_ERR_ARG;
    }
    // (16 workgroups a CU: 1e6 passwords are one a lane -- beside the sampler, where the bench scores its test set, the
    // step takes 0.596 ms against 0.600 with 8, 0.603 with 4, 0.612 with 2; 32 as 16)
    static const int env_swg = getenv("MC_SCORE_WG_PER_CU") ? atoi(getenv("MC_SCORE_WG_PER_CU")) : 16;
    unsigned long long want = (count + 255) / 256, cap = (unsigned long long)m->n_cu * (env_swg > 0 ? env_swg : 16);
    dim3 g((unsigned)(want < cap ? want : cap)), b(256);
    // passwords a lane walks at a time: one on the dense table (scripts/exp_score_grid.py, 1e6 passwords: the bench's
    // test set 0.040 ms against 0.051 with two, distinct strings 0.052 against 0.059), two on the hash path
    static const int env_np_given = getenv("MC_SCORE_CHAINS") ? atoi(getenv("MC_SCORE_CHAINS")) : 0;
    const int env_np = env_np_given > 0 ? env_np_given : (m->dev.dense ? 1 : 2);
    const unsigned long long *o64 = (const unsigned long long *)off;
    hipStream_t st = (hipStream_t)stream;
    if (m->dev.dense) {
        if (env_np == 1) hipLaunchKernelGGL((k_score<true, 1>), g, b, 0, st, m->dev, syms, o64, count, out_ml2p);
        else if (env_np == 4) hipLaunchKernelGGL((k_score<true, 4>), g, b, 0, st, m->dev, syms, o64, count, out_ml2p);
        else hipLaunchKernelGGL((k_score<true, 2>), g, b, 0, st, m->dev, syms, o64, count, out_ml2p);
    } else {
        if (env_np == 1) hipLaunchKernelGGL((k_score<false, 1>), g, b, 0, st, m->dev, syms, o64, count, out_ml2p);
        else hipLaunchKernelGGL((k_score<false, 2>), g, b, 0, st, m->dev, syms, o64, count, out_ml2p);
    }
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

static uint32_t words_lanes(const mc_model *m, uint64_t count) {
    unsigned long long want = ((count + 255) / 256) * 256, cap = (unsigned long long)m->n_cu * 8 * 256;
    return (uint32_t)(want < cap ? want : cap);
}

// k_score_words_fast takes passwords of at most WF_MAX characters (the end symbol's included) on a model whose symbols
// are at most WF_MAX long (MC_SCORE_WORDS_KERNEL=dfs: never); its workgroups per CU follow from its LDS
static std::atomic<int> g_words_kernel{0};
extern "C" void mc_score_words_debug_kernel(int which) { g_words_kernel.store(which); }

static bool words_fast_fits(const mc_model *m, uint32_t max_len) {
    static const bool never = getenv("MC_SCORE_WORDS_KERNEL") && !strcmp(getenv("MC_SCORE_WORDS_KERNEL"), "dfs");
    return !never && g_words_kernel.load() != 1 && m->dev.wordck && m->dev.max_word_len >= 1 && m->dev.max_word_len <= WF_MAX && (uint64_t)max_len + m->dev.end_text_len <= WF_MAX;
}

static size_t words_fast_lds(const mc_model *m, uint32_t max_len) {
    return ((size_t)max_len + m->dev.end_text_len + m->dev.max_word_len) * 256 * sizeof(uint32_t);
}

static uint32_t words_fast_lanes(const mc_model *m, uint64_t count, uint32_t max_len) {
    int per_cu = 0;
    const uint32_t w = m->dev.max_word_len;
    const void *fn = w <= 2 ? (const void *)k_score_words_fast<2> : w <= 4 ? (const void *)k_score_words_fast<4>
                   : w <= 6 ? (const void *)k_score_words_fast<6> : (const void *)k_score_words_fast<8>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, words_fast_lds(m, max_len)) != hipSuccess || per_cu < 1)
        per_cu = 1;
    unsigned long long want = ((count + 255) / 256) * 256, cap = (unsigned long long)m->n_cu * (unsigned)per_cu * 256;
    return (uint32_t)(want < cap ? want : cap);
}

extern "C" size_t mc_score_words_workspace_bytes(const mc_model *m, uint64_t count, uint32_t max_len) {
    if (!m || count == 0) return 0;
    // deepest recursion: every symbol one character, plus the end symbol, plus the outermost call
    size_t need = (size_t)words_lanes(m, count) * ((size_t)max_len + m->dev.end_text_len + 1) * sizeof(Frame);
    if (words_fast_fits(m, max_len)) {
        const size_t fast = (size_t)words_fast_lanes(m, count, max_len) * ((size_t)max_len + m->dev.end_text_len) * sizeof(WFrame);
        if (fast > need) need = fast;
    }
    return need;
}

extern "C" int mc_score_words(mc_model *m, const uint32_t *cps, const uint64_t *off, uint64_t count, uint32_t max_len,
                              uint64_t iter_max, double *out_ml2p, uint64_t *out_split, uint32_t split_words,
                              void *workspace, size_t workspace_bytes, void *stream) {
    if (!m || (count && (!off || !out_ml2p || !workspace)) || (out_split && split_words == 0)) {
        mc_set_error("mc_score_words: bad argument");
        return MC_ERR_ARG;
    }
    if (count == 0) return MC_OK;
    if (!m->has_text || (m->dev.kind != MC_KIND_NGRAM && m->dev.kind != MC_KIND_BACKOFF)) {
        mc_set_error("mc_score_words: needs an n-gram / backoff model compiled with symbol text "
                     "(mc_model_desc.sym_text_off)");
        return MC_ERR_ARG;
    }
    if (m->dev.end_text_len == 0 || max_len > 65000u) {
        mc_set_error("mc_score_words: empty end symbol or max_len above 65000");
        return MC_ERR_ARG;
    }
    if (words_fast_fits(m, max_len)) {
        const uint32_t rows = max_len + m->dev.end_text_len;
        const size_t per_lane = (size_t)rows * sizeof(WFrame);
        uint32_t lanes = words_fast_lanes(m, count, max_len);
        if ((size_t)lanes * per_lane > workspace_bytes) lanes = (uint32_t)((workspace_bytes / per_lane) / 256 * 256);
        if (lanes == 0) {
            mc_set_error("mc_score_words: workspace of %zu bytes cannot hold 256 DFS stacks of %zu bytes", workspace_bytes,
                         per_lane);
            return MC_ERR_WORKSPACE;
        }
        WordsArgs a{cps, (const unsigned long long *)off, count, iter_max, out_ml2p, (unsigned long long *)out_split,
                    (Frame *)workspace, split_words, max_len, lanes};
        const uint32_t w = m->dev.max_word_len;
        const dim3 g(lanes / 256), b(256);
        const size_t lds = words_fast_lds(m, max_len);
        if (w <= 2) hipLaunchKernelGGL(k_score_words_fast<2>, g, b, lds, (hipStream_t)stream, m->dev, a, rows);
        else if (w <= 4) hipLaunchKernelGGL(k_score_words_fast<4>, g, b, lds, (hipStream_t)stream, m->dev, a, rows);
        else if (w <= 6) hipLaunchKernelGGL(k_score_words_fast<6>, g, b, lds, (hipStream_t)stream, m->dev, a, rows);
        else hipLaunchKernelGGL(k_score_words_fast<8>, g, b, lds, (hipStream_t)stream, m->dev, a, rows);
        HIP_TRY(hipGetLastError());
        return MC_OK;
    }
    const size_t per_lane = ((size_t)max_len + m->dev.end_text_len + 1) * sizeof(Frame);
    uint32_t lanes = words_lanes(m, count);
    if ((size_t)lanes * per_lane > workspace_bytes) lanes = (uint32_t)((workspace_bytes / per_lane) / 256 * 256);
    if (lanes == 0) {
        mc_set_error("mc_score_words: workspace of %zu bytes cannot hold 256 DFS stacks of %zu bytes", workspace_bytes,
                     per_lane);
        return MC_ERR_WORKSPACE;
    }
    WordsArgs a{cps, (const unsigned long long *)off, count, iter_max, out_ml2p, (unsigned long long *)out_split,
                (Frame *)workspace, split_words, max_len, lanes};
    hipLaunchKernelGGL(k_score_words, dim3(lanes / 256), dim3(256), 0, (hipStream_t)stream, m->dev, a);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

static std::atomic<int> g_bpe_kernel{0};
extern "C" void mc_score_bpe_debug_kernel(int which) { g_bpe_kernel.store(which); }

extern "C" int mc_score_bpe(mc_model *m, const uint32_t *cps, const uint8_t *cls, const uint64_t *off, uint64_t count,
                            double *out_ml2p, void *stream) {
    if (!m || (count && (!off || !out_ml2p))) {
        mc_set_error("mc_score_bpe: bad argument");
        return MC_ERR_ARG;
    }
    if (count == 0) return MC_OK;
    if (m->dev.kind != MC_KIND_BPE) {
        mc_set_error("mc_score_bpe: not a BPE-PCFG model");
        return MC_ERR_ARG;
    }
    static const bool walk_env = getenv("MC_SCORE_BPE_KERNEL") && !strcmp(getenv("MC_SCORE_BPE_KERNEL"), "walk");
    const bool fast = !walk_env && g_bpe_kernel.load() != 1 && m->dev.wordck;
    // (k_score_bpe_fast: 32 KB of LDS and 113 VGPRs: four workgroups share a CU)
    unsigned long long want = (count + 255) / 256, cap = (unsigned long long)m->n_cu * (fast ? 4 : 8);
    dim3 g((unsigned)(want < cap ? want : cap)), b(256);
    if (fast)
        hipLaunchKernelGGL(k_score_bpe_fast, g, b, 0, (hipStream_t)stream, m->dev, cps, cls, (const unsigned long long *)off,
                           (unsigned long long)count, out_ml2p);
    else
        hipLaunchKernelGGL(k_score_bpe, g, b, 0, (hipStream_t)stream, m->dev, cps, cls, (const unsigned long long *)off,
                           (unsigned long long)count, out_ml2p);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

extern "C" int mc_sample_dedup(const uint32_t *syms, const uint32_t *len, uint32_t sym_stride, uint64_t count,
                               uint64_t cap, uint64_t *tab_key, uint64_t *tab_first, uint32_t *tab_cnt,
                               uint32_t *mismatch, void *stream) {
    if (!tab_key || !tab_first || !tab_cnt || !mismatch || (count && (!syms || !len)) || sym_stride == 0 ||
        cap < 2 * count || cap == 0 || (cap & (cap - 1))) {
        mc_set_error("mc_sample_dedup: bad argument (cap must be a power of two >= 2 * count)");
        return MC_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(tab_key, 0xFF, cap * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(tab_first, 0xFF, cap * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(tab_cnt, 0, cap * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(mismatch, 0, sizeof(uint32_t), s));
    if (count == 0) return MC_OK;
    unsigned long long want = (count + 255) / 256;
    dim3 g((unsigned)(want < 8192 ? want : 8192)), b(256);
    hipLaunchKernelGGL(k_dedup_insert, g, b, 0, s, syms, len, sym_stride, (unsigned long long)count,
                       (unsigned long long)(cap - 1), (unsigned long long *)tab_key, (unsigned long long *)tab_first,
                       tab_cnt);
    hipLaunchKernelGGL(k_dedup_verify, g, b, 0, s, syms, len, sym_stride, (unsigned long long)count,
                       (unsigned long long)(cap - 1), (const unsigned long long *)tab_key,
                       (const unsigned long long *)tab_first, mismatch);
    HIP_TRY(hipGetLastError());
    return MC_OK;
}
