// Internal declarations shared by the table compiler (host) and the HIP side.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mi355mc.h"

#define MC_STATUS_WORDS 8
// The counters are kept as MC_STATUS_SLOTS copies, one 128-byte line each: a wave adds to the copy its
// (workgroup, wave) number selects and mc_model_check sums the copies.  Thousands of atomics on ONE
// address serialise at the memory side (measured: 16k of them cost 0.18 ms, 3/4 of k_score's time).
#define MC_STATUS_SLOTS 256
#define MC_STATUS_STRIDE 16
// status/counter words kept in device memory per model
enum {
    MC_CNT_DRAWS = 0,      // uniforms consumed
    MC_CNT_PROBES = 1,     // cum[] comparisons inside guide buckets
    MC_CNT_RESTARTS = 2,   // samples restarted (too short / too long)
    MC_CNT_FAILED = 3,     // samples that hit max_draws or a missing row
    MC_CNT_SCORE_PROBES = 4,  // transition-hash slots read while scoring
    MC_CNT_SCORE_STEPS = 5,   // symbols consumed while scoring
    MC_CNT_INDIRECT = 6,      // draws whose bucket was not DIRECT (one more 16-byte record read)
    MC_CNT_SCORE_TOO_LONG = 7 // passwords longer than mc_score_words' max_len
};

#define MC_CK_POS1(h) (h)
#define MC_CK_POS2(h) ((((h) ^ ((h) >> 29)) * 0x9E3779B97F4A7C15ull) >> 21)
#define MC_CK_TAG(h, n) ((uint32_t)((h) >> 32) ^ ((uint32_t)(n) * 0x9E3779B1u))

struct mc_table {
    uint32_t kind, order, n_ctx, n_sym;
    uint64_t n_ent;
    uint32_t end_sym, min_len, max_len, flags;
    uint32_t start_state, root, max_row;
    bool multichar;
    bool len_in_rec;  // the draw records carry the symbols' lengths (MC_DRAW_LEN_SHIFT)
    std::vector<mc_state_rec> state;
    std::vector<mc_draw_rec> draw;   // bucketed draw table (all rows)
    std::vector<mc_draw_rec> spay;   // per-entry DIRECT records
    std::vector<double> rcum;        // per-entry thresholds T_j
    uint32_t start_draw, start_gbits;
    uint64_t n_direct, n_one, n_multi;
    std::vector<double> cum;
    std::vector<mc_payload> payload;
    std::vector<mc_slot> trans;
    std::vector<mc_slot> ctxhash;
    std::vector<uint32_t> sym_chars;
    std::vector<mc_payload> dense;        // [n_ctx][n_sym] resolved transitions (small single-character alphabets)
    std::vector<mc_slot> wordhash;        // empty when the descriptor had no symbol text
    // the same map as a two-choice table of 8-byte slots (tag << 32 | symbol) for look-ups that must end in ONE round
    // of loads (k_score_words_fast): a text's slot is MC_CK_POS1 or MC_CK_POS2 of its hash; a tag that matches is
    // checked against the symbol's text by the caller.  Empty slots are ~0; empty vector: none could be built
    std::vector<uint64_t> wordck;
    std::vector<uint32_t> sym_text_off, sym_text;
    uint32_t max_word_len, start_sym;
    // MC_KIND_BPE
    std::vector<uint32_t> seg_off, seg_row, seg_len;
    std::vector<mc_slot> labhash;  // key = mc_ctx_hash(label units), val = label index
    std::vector<uint32_t> lab_off, lab_units, lab_cand_off, lab_cand, np_len, np_off, np_cand;
};

void mc_set_error(const char *fmt, ...);
uint64_t mc_trans_hash(uint64_t key);
size_t mc_table_device_bytes(const mc_table *t);

// what the enumeration kernels (mc_enum.hip) read of a resident model (defined in mc_device.hip)
struct mc_enum_view {
    const mc_state_rec *state;
    const mc_payload *payload;
    const uint32_t *sym_chars;
    uint32_t start_state, root, n_ctx, n_sym, end_sym, kind;
    int device;
};
int mc_model_enum_view(const mc_model *m, mc_enum_view *out);
