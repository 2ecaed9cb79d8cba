/*
 * mi355mc.h -- C ABI of libmi355mc.so, the MI355X (gfx950) Monte Carlo
 * guess-number engine behind kofny/MonteCarlo's lib4mc plugin surface.
 *
 * Plain pointers and sizes only.  "device" pointers are HIP device memory of
 * the GPU that is current on the calling thread (e.g. torch tensors'
 * data_ptr()); "host" pointers are ordinary memory.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  Every call returns MC_OK (0) or
 * a negative mc_status; mc_last_error() gives the text for the calling thread.
 * Launch calls are asynchronous on `stream`; none of them allocates or frees
 * device memory (the caller brings outputs and workspaces).
 *
 * Each entry point names the reference interface (file:line under
 * /root/reference) whose inside it replaces.  INTEGRATION.md shows the ctypes
 * stubs a maintainer adds to the reference to bind them.
 */
#ifndef MI355MC_H
#define MI355MC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 2  /* 2: mc_format_gc_cps takes the number of unique lines; the run-length rank index */

typedef enum mc_status {
    MC_OK = 0,
    MC_ERR_ARG = -1,          /* bad argument (null pointer, size mismatch, unknown kind) */
    MC_ERR_TABLE = -2,        /* table violates an assumption the compiler checks (message says which) */
    MC_ERR_HIP = -3,          /* a HIP runtime call failed */
    MC_ERR_NOMEM = -4,
    MC_ERR_DIVERGED = -5,     /* a sample hit max_draws without reaching the end symbol */
    MC_ERR_WORKSPACE = -6,    /* workspace too small, or a build failed on the device (mc_rank_check) */
    MC_ERR_UNSUITED = -7      /* mc_rank_runs_check: the keys do not suit the run index; use mc_rank_build */
} mc_status;

/* Context rule of the model (which `_get_prefix` it restates). */
enum {
    MC_KIND_NGRAM = 1,   /* nwords_simulator.py:30-34: context = last `order` symbols (exact row required) */
    MC_KIND_BACKOFF = 2, /* backwords_simulator.py:28-41: longest suffix of the history that is a row */
    MC_KIND_V2 = 3,      /* nwords_simulator_v2.py:65-68: context = last `order` CHARACTERS; transitions are words */
    MC_KIND_BPE = 4      /* bpe_simulator.py:104-157: row 0 = the grammar (entry i = structure i), rows 1.. = the
                          * terminal rows; every row has the one-unit context {row index}; start_units = {0} */
};

enum {
    MC_FLAG_KEEP_LEN_ON_RESTART = 1u, /* nwords_simulator_v2.py:73-81: pwd_len is not reset on restart */
    MC_FLAG_NO_DENSE = 2u             /* do not build the dense (row, symbol) scoring table (MC_ARR_DENSE) */
};

#define MC_NONE 0xFFFFFFFFu  /* "no row" / unknown symbol id */

/*
 * A trained model in the reference's own terms, flattened by the caller:
 * what `expand_2d(trainer(...))` holds (lib4mc/ProbLib.py:10-28), row by row in
 * dict order.  Row r has context units ctx_units[ctx_off[r] .. ctx_off[r+1])
 * and entries [row_off[r], row_off[r+1]); entry e is transition symbol
 * ent_sym[e] with ent_cum[e] = the row's sequential fp64 cumulative sum up to
 * and including e (exactly numpy's cumsum, ProbLib.py:23) and ent_ml2p[e] =
 * -log2(p) as the reference's minus_log2 computes it (MonteCarloParent.py:13).
 * All pointers are host memory and are only read during mc_table_compile.
 */
typedef struct mc_model_desc {
    uint32_t kind;          /* MC_KIND_* */
    uint32_t order;         /* n-1 for NGRAM / V2; ignored for BACKOFF */
    uint32_t n_ctx;
    uint32_t n_sym;         /* number of transition symbol ids */
    uint64_t n_ent;
    const uint32_t *ctx_off;      /* [n_ctx+1] */
    const uint32_t *ctx_units;    /* [ctx_off[n_ctx]] */
    const uint64_t *row_off;      /* [n_ctx+1] */
    const uint32_t *ent_sym;      /* [n_ent] */
    const double *ent_cum;        /* [n_ent] */
    const double *ent_ml2p;       /* [n_ent] */
    const uint32_t *sym_unit_off; /* [n_sym+1] or NULL: context units a symbol appends (NULL: the symbol itself) */
    const uint32_t *sym_units;
    const uint32_t *sym_chars;    /* [n_sym] len(symbol) in characters, or NULL: all 1 */
    const uint32_t *start_units;  /* initial history (n-1 start symbols; one for BACKOFF; none for V2) */
    uint32_t n_start;
    uint32_t end_sym;       /* transition symbol id of end_chr */
    uint32_t min_len;       /* accept a sample iff pwd_len >= min_len (nwords_simulator.py:101; v2 `> 3`) */
    uint32_t max_len;       /* restart when pwd_len >= max_len (nwords_simulator.py:112): 256 */
    uint32_t flags;         /* MC_FLAG_* */
    uint32_t guide_factor;  /* draw-table budget: ~guide_factor * n_ent buckets, shared out by expected visits (0 -> 2) */
    /* text of every transition symbol as Unicode code points (symbol s = sym_text[sym_text_off[s] ..
     * sym_text_off[s+1])): what `left in self.words` / `left in row` compare against in _structures
     * (nwords_simulator.py:42-45).  NULL: mc_score_words is unavailable for the model. */
    const uint32_t *sym_text_off; /* [n_sym+1] or NULL */
    const uint32_t *sym_text;
    /* ---- MC_KIND_BPE only (NULL / 0 otherwise) ------------------------------------------------
     * Structure i is the segments seg_off[i] .. seg_off[i+1]: seg_row[k] = the row of terminals
     * (tag, t_len) (MC_NONE when the model has none: the reference raises KeyError there), seg_len[k] =
     * t_len.  calc_ml2p's candidate index (count_luds, bpe_simulator.py:45-101): label j is the units
     * lab_units[lab_off[j] .. lab_off[j+1]) (unit = class << 24 | run length; class 0..3 = L U D S, luds
     * :17-40) with candidate structures lab_cand[lab_cand_off[j] .. lab_cand_off[j+1]); not_parsed[k] is
     * the structures np_cand[np_off[k] .. np_off[k+1]) of total length np_len[k]. */
    const uint32_t *seg_off, *seg_row, *seg_len;
    uint32_t n_labels, n_np;
    const uint32_t *lab_off, *lab_units, *lab_cand_off, *lab_cand;
    const uint32_t *np_len, *np_off, *np_cand;
} mc_model_desc;

typedef struct mc_table mc_table; /* compiled table, host memory */
typedef struct mc_model mc_model; /* compiled table resident on one GPU */

int mc_abi_version(void);
const char *mc_last_error(void);

/* ---- table compiler (host only; runs without a GPU) --------------------------------------
 * Replaces the per-call dict/tuple work of `_get_prefix` (nwords_simulator.py:30-34,
 * backwords_simulator.py:28-41) and `pick_expand`'s bisect (lib4mc/ProbLib.py:31-43) by
 * precomputed arrays: per-entry next-state links, per-row parent (backoff) links, an exact
 * bucketed draw table for the inverse-CDF draw (see mc_draw_rec), an open-addressed
 * (row,symbol)->entry hash and an open-addressed 64-bit context-hash -> row index.
 */
int mc_table_compile(const mc_model_desc *desc, mc_table **out);
void mc_table_free(mc_table *t);

/* Introspection of a compiled table (used by tests and by the host classes): copies of the
 * flat arrays.  `what` names the array; returns its byte size, copies at most `cap` bytes. */
enum {
    MC_ARR_STATE = 0,    /* mc_state_rec[n_ctx] */
    MC_ARR_DRAW = 1,     /* mc_draw_rec[]: per row 2^gbits buckets over the uniform */
    MC_ARR_CUM = 2,      /* double[n_ent]: the reference's cumsum (host only; the device uses RCUM) */
    MC_ARR_PAYLOAD = 3,  /* mc_payload[n_ent] (scoring) */
    MC_ARR_TRANS = 4,    /* mc_slot[trans_cap]: key = row<<32|sym, val = entry */
    MC_ARR_CTXHASH = 5,  /* mc_slot[ctx_cap]: key = 64-bit context hash, val = row */
    MC_ARR_SYMCHARS = 6, /* uint32[n_sym] */
    MC_ARR_SPAY = 7,     /* mc_draw_rec[n_ent]: MC_DRAW_DIRECT record of every entry (sampling) */
    MC_ARR_RCUM = 8,     /* double[n_ent]: per-entry thresholds in the uniform's domain (see mc_draw_rec) */
    MC_ARR_WORDHASH = 9, /* mc_slot[word_cap]: key = mc_text_hash(symbol text), val = symbol id, pad = length */
    MC_ARR_SYMTEXTOFF = 10, /* uint32[n_sym+1] */
    MC_ARR_SYMTEXT = 11, /* uint32[]: code points */
    MC_ARR_DENSE = 12,   /* mc_payload[n_ctx * n_sym] or empty: for single-character alphabets small enough
                          * (n_ctx * n_sym <= 2^22), the transition (row, symbol) with the backoff already
                          * resolved: {ml2p, next, entry} of the longest suffix row that lists the symbol
                          * (backwords_simulator.py:31-38), entry = MC_NONE when there is none */
    MC_ARR_WORDCK = 13   /* uint64[ck_cap]: the symbol-text map of MC_ARR_WORDHASH once more as a two-choice table for
                          * look-ups that must end in one round of loads (k_score_words_fast, k_score_bpe_fast): a
                          * symbol s with text hash h = mc_text_hash(text) and length n sits in slot h & (ck_cap - 1) or
                          * in slot (((h ^ h >> 29) * 0x9E3779B97F4A7C15) >> 21) & (ck_cap - 1) as
                          * ((uint32)(h >> 32) ^ n * 0x9E3779B1) << 32 | s; empty slots are all ones.  A tag that
                          * matches is checked against the symbol's text by the reader.  Empty: no such table */
};
int64_t mc_table_array(const mc_table *t, int what, void *dst, size_t cap);
/* start state, end symbol, root (zero-gram) row, table sizes ... as 16 uint64 (see mc_table_info_idx) */
int mc_table_info(const mc_table *t, uint64_t info[16]);
enum {
    MC_INFO_NCTX = 0, MC_INFO_NENT = 1, MC_INFO_NSYM = 2, MC_INFO_START_STATE = 3, MC_INFO_ROOT = 4,
    MC_INFO_TRANS_CAP = 5, MC_INFO_CTX_CAP = 6, MC_INFO_GUIDE_LEN = 7, MC_INFO_DEVICE_BYTES = 8,
    MC_INFO_MAX_ROW = 9, MC_INFO_KIND = 10, MC_INFO_MULTICHAR = 11, MC_INFO_START_DRAW = 12,
    MC_INFO_DIRECT_BUCKETS = 13, MC_INFO_ONE_BUCKETS = 14, MC_INFO_MULTI_BUCKETS = 15
};  /* info[11] (MC_INFO_MULTICHAR) bit 0: some symbol is not one character; bit 1: the draw records carry the
     symbols' lengths (MC_DRAW_LEN_SHIFT); bits 8..: longest symbol text */
/* 64-bit hash of a context (sequence of unit ids) as stored in the context index. */
uint64_t mc_ctx_hash(const uint32_t *units, uint32_t n);
/* 64-bit hash of a symbol's text (code points) as stored in the word index. */
uint64_t mc_text_hash(const uint32_t *cps, uint32_t n);

/* Flat record layouts (also the device layouts). */
typedef struct mc_state_rec {
    double total;       /* cum[last] of the row: random.uniform(0, total), ProbLib.py:40-41 */
    uint32_t ent_off;   /* first entry of the row */
    uint32_t len;       /* entries in the row */
    uint32_t guide_off; /* first draw-table bucket (mc_draw_rec index) of the row */
    uint32_t gbits;     /* log2(buckets of the row) */
    uint32_t parent;    /* longest proper suffix that is a row (BACKOFF), else MC_NONE */
    uint32_t ctx_len;   /* context length in units */
} mc_state_rec;

typedef struct mc_payload {
    double ml2p;        /* -log2 p of the entry */
    uint32_t next;      /* state after taking the entry (MC_NONE: no such row) */
    uint32_t sym;       /* transition symbol id; bit 31 set on the end symbol */
} mc_payload;

/*
 * Draw table: the reference draws idx = bisect_right(cum, total * r) with r = random.random()
 * (lib4mc/ProbLib.py:40-42).  fl(total * r) is monotone in r, so for every entry j of a row
 * there is a smallest double T_j in [0,1] with fl(total * T_j) >= cum[j]; then
 *     idx = #{ j < len-1 : r >= T_j }                        (same index, bit for bit)
 * and the device never multiplies: RCUM[j] = T_j (the row's last entry holds 2.0: the reference's
 * IndexError at u == cum[-1] is clamped to the last entry).  A row owns 2^gbits buckets; bucket
 * k = floor(r * 2^gbits) is ONE 16-byte record telling what r in [k, k+1) / 2^gbits lands on:
 *   MC_DRAW_DIRECT  every such r lands on one entry: the record IS that entry's sampling
 *                   payload {ml2p, next, meta};
 *   MC_DRAW_ONE     one threshold inside the bucket: {T, lo}; idx = lo + (r >= T);
 *   MC_DRAW_MULTI   cnt >= 2 thresholds: {cnt, -, lo}; idx = lo + bisect_right(RCUM[lo..lo+cnt), r).
 * For ONE / MULTI the entry's record is then read from SPAY[idx].
 * meta (word 3): bit 31 end symbol, bits 30-29 record type, bits 28-24 gbits of the NEXT row,
 * bits 23-0 symbol id; `next` (word 2) = first bucket of the NEXT row (MC_NONE: no such row),
 * so a draw needs nothing but its own record to continue.
 */
enum { MC_DRAW_DIRECT = 0, MC_DRAW_ONE = 1, MC_DRAW_MULTI = 2 };
#define MC_DRAW_END 0x80000000u
#define MC_DRAW_TYPE(meta) (((meta) >> 29) & 3u)
#define MC_DRAW_GBITS(meta) (((meta) >> 24) & 31u)
#define MC_DRAW_SYM(meta) ((meta) & 0x00FFFFFFu)
/* Models whose symbols are not all one character (info[MC_INFO_MULTICHAR] bit 0) and whose draw table has fewer
 * than 2^26 - 1 buckets keep the symbol's length beside the next row (info[MC_INFO_MULTICHAR] bit 1): word 2 of a
 * DIRECT record that is not the end symbol's = characters of the symbol << 26 | the next row's first bucket
 * (MC_DRAW_OFF_MASK: no next row), so the sampler's length count needs no read of its own. */
#define MC_DRAW_LEN_SHIFT 26
#define MC_DRAW_OFF_MASK 0x03FFFFFFu
#define MC_DRAW_LEN_MAX 63u
#define MC_MAX_SYMBOLS 0x01000000u /* symbol ids must fit the 24-bit field */

typedef struct mc_draw_rec {
    uint32_t w[4];      /* DIRECT: f64 ml2p | next | meta;  ONE: f64 T | lo | meta;  MULTI: cnt | 0 | lo | meta */
} mc_draw_rec;

typedef struct mc_slot {
    uint64_t key;       /* ~0 = empty */
    uint32_t val;
    uint32_t pad;
} mc_slot;

/* ---- trainers (host only): nwords_counter (nwords/nwords_trainer.py:22-53) and backwords_counter
 * (backwords/backwords_trainer.py:23-78) on the decoded text of the training file (UTF-8, '\n' line
 * ends), with the reference's dict orders and fp64 arithmetic, straight into flattened rows.
 * splitter_cp = 0: every character is a symbol (splitter ''); otherwise the one-character column
 * separator (' ', '\t'), columns taken start4word::skip4word.  start_cp / end_cp: start_chr / end_chr. */
typedef struct mc_trained mc_trained;
int mc_train_nwords(const char *utf8, size_t bytes, uint32_t n, uint32_t splitter_cp, uint32_t start4word,
                    uint32_t skip4word, uint32_t start_cp, uint32_t end_cp, mc_trained **out);
int mc_train_backwords(const char *utf8, size_t bytes, uint32_t splitter_cp, uint32_t start4word, uint32_t skip4word,
                       uint32_t threshold, uint32_t max_gram, uint32_t start_cp, uint32_t end_cp, mc_trained **out);
/* nwords_counter of the v2 model (nwords_v2/nwords_trainer_v2.py:12-56): lines "pwd\tseg\ttag\tseg\ttag...",
 * contexts = the last n-1 CHARACTERS, transitions = pieces seen at least `threshold` times, else their
 * characters.  MC_TRAINED_CTX_UNITS of such a model holds code points (not symbol ids); it compiles as
 * MC_KIND_V2 only.  A line whose segments do not rebuild the password is MC_ERR_ARG ("error1" in the reference). */
int mc_train_v2(const char *utf8, size_t bytes, uint32_t n, uint32_t end_cp, uint32_t threshold, mc_trained **out);
void mc_trained_free(mc_trained *t);
enum {
    MC_TRAINED_CTX_OFF = 0,      /* uint32[n_ctx+1] */
    MC_TRAINED_CTX_UNITS = 1,    /* uint32[]: symbol ids of every row's context */
    MC_TRAINED_ROW_OFF = 2,      /* uint64[n_ctx+1] */
    MC_TRAINED_ENT_SYM = 3,      /* uint32[n_ent] */
    MC_TRAINED_ENT_PROB = 4,     /* double[n_ent]: the trainer's {symbol: p} values, row by row in dict order */
    MC_TRAINED_ENT_CUM = 5,      /* double[n_ent]: expand_1d's cumulative sums (lib4mc/ProbLib.py:23) */
    MC_TRAINED_ENT_ML2P = 6,     /* double[n_ent]: -log2 p */
    MC_TRAINED_SYM_TEXT_OFF = 7, /* uint32[n_sym+1] */
    MC_TRAINED_SYM_TEXT = 8,     /* uint32[]: code points of every symbol */
    MC_TRAINED_WORD_SYM = 9,     /* uint32[]: the trainer's `words` dict, symbols in first-seen order ... */
    MC_TRAINED_WORD_COUNT = 10   /* uint64[]: ... and their occurrence counts */
};
int64_t mc_trained_array(const mc_trained *t, int what, void *dst, size_t cap);

/* Secondary (incremental) training, backwords/backwords_secondary_trainer.py: the raw counts persist
 * between rounds.  mc_counts_merge = backwords_counter (:23-99) on one round's training text (a new
 * context is admitted only with a transition at/above the threshold); mc_counts_train = freq2prob
 * (:102-121).  mc_counts_array copies the counts out (`what` as mc_trained_array; the row of the empty
 * context comes first, then the rows in the insertion order of nwords_dict; MC_COUNTS_ENT_COUNT =
 * uint64[n_ent] replaces the probability arrays). */
typedef struct mc_counts mc_counts;
#define MC_COUNTS_ENT_COUNT 16
int mc_counts_new(mc_counts **out);
void mc_counts_free(mc_counts *k);
int mc_counts_merge(mc_counts *k, const char *utf8, size_t bytes, uint32_t splitter_cp, uint32_t start4word,
                    uint32_t skip4word, uint32_t max_gram, uint32_t threshold, uint32_t start_cp, uint32_t end_cp);
int mc_counts_train(const mc_counts *k, uint32_t threshold, mc_trained **out);
/* Counts that exist as plain arrays -- the `(counts, words, config)` pickle the reference writes per round
 * (backwords_secondary_main.py:28-32, 215-217) and reads back (backwords_secondary_trainer.py:36-42,
 * backwords_secondary_simulator.py:17) -- into an EMPTY mc_counts, so that resumed rounds merge and train
 * natively.  Symbols (texts as code points) keep the order given, `words` and the rows keep theirs (dict
 * insertion order); the row of the empty context is implied by `words` and is not passed. */
int mc_counts_import(mc_counts *k, const uint32_t *sym_text_off, const uint32_t *sym_text, uint32_t n_sym,
                     const uint32_t *word_sym, const uint64_t *word_cnt, uint32_t n_words, const uint32_t *ctx_off,
                     const uint32_t *ctx_units, const uint64_t *row_off, const uint32_t *ent_sym,
                     const uint64_t *ent_cnt, uint32_t n_rows, uint32_t start_cp, uint32_t end_cp);
int64_t mc_counts_array(const mc_counts *k, int what, void *dst, size_t cap);
/* mc_table_compile on the trained rows (kind MC_KIND_NGRAM with order = n-1, or MC_KIND_BACKOFF). */
int mc_trained_compile(const mc_trained *t, uint32_t kind, uint32_t order, uint32_t min_len, uint32_t max_len,
                       uint32_t flags, uint32_t guide_factor, mc_table **out);

/* ---- test-file ingest (host) ---------------------------------------------------------------*/
/* Replaces the counting loop of MonteCarlo.parse_file (lib4mc/MonteCarloParent.py:56-60): the lines of
 * `utf8` ('\n' line ends; '\r' and '\n' stripped from both ends, as line.strip("\r\n")), deduplicated in
 * first-seen order -- the insertion order of the reference's pwd_counter dict.  mc_lines_array copies
 * up to `cap` bytes into dst and returns the full size in bytes (as mc_trained_array does). */
typedef struct mc_lines mc_lines;
enum {
    MC_LINES_COUNT = 0, /* uint64[n_unique]    occurrences of each unique line                         */
    MC_LINES_OFF = 1,   /* uint64[n_unique+1]  offsets of the unique lines in MC_LINES_CPS             */
    MC_LINES_CPS = 2,   /* uint32[]            their code points, concatenated (mc_score_words' input) */
    MC_LINES_INFO = 3   /* uint64[4]           unique lines, lines, longest unique line, code points   */
};
int mc_lines_ingest(const char *utf8, size_t bytes, mc_lines **out);
void mc_lines_free(mc_lines *l);
int64_t mc_lines_array(const mc_lines *l, int what, void *dst, size_t cap);
/* Replaces the formatting loop of MonteCarloLib.write2 (lib4mc/MonteCarloLib.py:65-68): one line
 * "pwd\t%.8f\tappearance\trank\tcracked\t%5.2f\n" per password, the passwords given as one '\n'-separated
 * UTF-8 text.  Returns the bytes written to dst (len(text) + 120 * n is always enough), negative on error. */
int64_t mc_format_gc(const char *pwds_utf8, size_t bytes, uint64_t n, const double *mlp, const int64_t *appearance,
                     const int64_t *rank, const int64_t *cracked, const double *ratio, char *dst, size_t cap);

/* The same lines with the passwords as the code points of mc_lines_ingest (MC_LINES_CPS, MC_LINES_OFF: n_lines unique
 * lines, off[n_lines + 1]) and an order: row i is unique line order[i] (n rows; an order[i] >= n_lines is MC_ERR_ARG,
 * nothing is read through it).  len(cps) * 4 + 120 * n bytes are always enough.  A lone surrogate among the code
 * points has no UTF-8 form: MC_ERR_ARG (the caller writes such a file with the reference's loop).  Runs on as many
 * threads as it can start (at most 16, at least the caller's own); MC_ERR_NOMEM when memory runs out. */
int64_t mc_format_gc_cps(const uint32_t *cps, const uint64_t *off, uint64_t n_lines, const uint32_t *order, uint64_t n,
                         const double *mlp, const int64_t *appearance, const int64_t *rank, const int64_t *cracked,
                         const double *ratio, char *dst, size_t cap);

/* ---- device residency ---------------------------------------------------------------------*/
/* Uploads the compiled table to GPU `device` (one hipMalloc arena). */
int mc_model_upload(const mc_table *t, int device, mc_model **out);
void mc_model_destroy(mc_model *m);
/* Synchronises `stream`, reads and clears the model's device status words.  Returns
 * MC_ERR_DIVERGED if any sample since the last check exhausted its draws.  counters (may be
 * NULL) receives {draws, threshold compares, restarts, failed samples, score probes, score steps,
 * indirect draws (bucket not DIRECT), -} (8 words). */
int mc_model_check(mc_model *m, void *stream, uint64_t counters[8]);

/* ---- sampling: MonteCarlo.sample, lib4mc/MonteCarloParent.py:33-46, whose loop body is
 * sample1 (nwords_simulator.py:92-116 / nwords_simulator_v2.py:60-82) + pick_expand
 * (lib4mc/ProbLib.py:31-43).  Sample ids [first_id, first_id+count) use the Philox4x32-10
 * stream (seed, id, draw) -> 53-bit uniform defined in oracle/philox.py; results do not
 * depend on launch geometry.  Outputs (device): out_ml2p[count]; optional out_len[count]
 * (symbols emitted), out_draws[count] (uniforms consumed, restarts included), and
 * out_syms[count*sym_stride] (symbol ids; samples longer than sym_stride are truncated
 * there, out_len still reports the full length).  For MC_KIND_BPE the walk is BpePcfgSim.sample1
 * (bpe_simulator.py:105-116): one structure draw, then one terminal draw per segment, no
 * rejection; out_syms then holds the ENTRY index of every terminal drawn and out_len the number
 * of segments. */
int mc_sample(mc_model *m, uint64_t first_id, uint64_t count, uint64_t seed, uint32_t max_draws,
              double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
              uint32_t sym_stride, void *stream);

/* mc_sample that takes a copy along: carry_bytes bytes from carry_src (device) to carry_dst (pinned host memory as
 * a rule; any address the GPU can write) are moved by a few workgroups of the sampling launch itself, beside the
 * sampling.  No counterpart in the reference: it is for pipelined callers (bench.py's step) whose previous
 * results must reach the host while the next samples are drawn -- on this GPU a copy on a second stream slows the
 * sampler by a quarter, a copy in the same stream waits for it; inside the launch it costs nothing.  Samples are
 * those of mc_sample bit for bit.  Both addresses 16-byte aligned, carry_bytes a multiple of 16 (else MC_ERR_ARG);
 * carry_bytes 0 is mc_sample; with count 0 or an MC_KIND_BPE model the copy is the runtime's, on `stream`.  The
 * destination is complete when the launch is (stream order), not before. */
int mc_sample_carry(mc_model *m, uint64_t first_id, uint64_t count, uint64_t seed, uint32_t max_draws,
                    double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
                    uint32_t sym_stride, const void *carry_src, void *carry_dst, uint64_t carry_bytes, void *stream);

/* Launch geometry of the sampling kernel for this model (no counterpart in the reference; samples never depend on
 * it): workgroups per compute unit (default 14, of which 8 are resident: best when the sampler has the GPU to
 * itself, and 7 for mc_sample_carry; 7 leaves every compute unit a free slot for the kernels of a second stream -- bench.py scores and orders
 * the test set there), the skew of the workgroups' id ranges (default 0.55: ranges shrink linearly from 1.55 x to
 * 0.45 x the mean) and the KB of the draw table's head kept in LDS (default 15).  wg_per_cu <= 0, skew < 0 or
 * hot_kb < 0 put the default back (the MC_SAMPLE_* environment variables set the defaults).  Not synchronised
 * with launches in flight on other threads. */
int mc_sample_tune(mc_model *m, int wg_per_cu, float skew, int hot_kb);

/* ---- sampled strings, deduplicated on the GPU: what MonteCarlo.sample builds in `sampled_pwds`
 * (lib4mc/MonteCarloParent.py:40-45: string -> [ml2p of its first occurrence, occurrences], in
 * first-sampled order) and backwords_secondary_main.py:45-51 consumes.  Input: the symbol rows
 * mc_sample wrote (out_syms, out_len, sym_stride).  Every string is inserted into an open-addressed
 * table of `cap` slots (a power of two >= 2 * count; the three table arrays are outputs the call
 * initialises): tab_key = 64-bit hash of the row (~0 = empty slot), tab_first = smallest sample
 * index holding the string, tab_cnt = occurrences.  Rows longer than sym_stride are skipped
 * (tab_cnt does not count them: the caller replays those sample ids).  A second kernel compares
 * every row with its slot's first row; *mismatch (device) counts the rows that differ, i.e. hash
 * collisions -- the caller must treat a non-zero count as an error. */
int mc_sample_dedup(const uint32_t *syms, const uint32_t *len, uint32_t sym_stride, uint64_t count, uint64_t cap,
                    uint64_t *tab_key, uint64_t *tab_first, uint32_t *tab_cnt, uint32_t *mismatch, void *stream);

/* Same walk with caller-injected uniforms (parity mode: "same injected uniform stream ->
 * bit-identical strings").  Sample i consumes uniforms[i*draws_per_sample + k], k = 0,1,...;
 * a sample that needs more than draws_per_sample is reported failed (ml2p = NaN). */
int mc_sample_injected(mc_model *m, uint64_t count, const double *uniforms, uint32_t draws_per_sample,
                       double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
                       uint32_t sym_stride, void *stream);

/* ---- scoring: calc_ml2p over a batch (MonteCarloParent.py:62-67 calling
 * nwords_simulator.py:36-60,84-90 / backwords_simulator.py:43-49), single-symbol alphabets
 * (every word is one character: exactly one parse).  Password t is the symbol ids
 * syms[off[t] .. off[t+1]) (MC_NONE for a character outside the model); the end symbol is
 * appended by the kernel.  out_ml2p[t] = sum of -log2 p in order, or 1022.0 when there is
 * no parse (nwords_simulator.py:85). */
int mc_score(mc_model *m, const uint32_t *syms, const uint64_t *off, uint64_t count, double *out_ml2p,
             void *stream);

/* ---- scoring, any alphabet: calc_ml2p as the reference's DFS over every segmentation of
 * pwd + end_chr into symbols (`_structures`, nwords_simulator.py:36-60, called from :84-90 and
 * backwords_simulator.py:43-49): the MINIMUM over full parses of the left-to-right sum of -log2 p,
 * exploring split lengths in ascending order and honouring iter_max exactly as the recursion does
 * (a callee returns at once when iter_max full parses have been seen; callers go on with their
 * remaining splits).  Password t is the code points cps[off[t] .. off[t+1]); the end symbol's text
 * is appended by the kernel.  out_ml2p[t] = that minimum, or 1022.0 when there is no parse.
 * out_split (may be NULL) receives split_words u64 per password: bit j set = a symbol of the best
 * parse ends after character j+1 of pwd + end_chr (all zero when unparseable).  max_len = an upper
 * bound on the password lengths of the batch (the DFS stack lives in `workspace`); a longer
 * password yields NaN and is reported by mc_model_check as a failed item.
 * Cost: the reference's -- EVERY segmentation is visited.  Where the model's symbols include every short string over
 * a small alphabet, a password of L characters has about 1.9^L of them (3e8 at L = 30): the launch then runs for as
 * long as that walk takes and cannot be interrupted.  iter_max is the bound the reference offers, and the one to
 * use for such models (scripts/model_fuzz.py met the case: bigrams over tokens of 1 .. 4 characters from 2 letters). */
size_t mc_score_words_workspace_bytes(const mc_model *m, uint64_t count, uint32_t max_len);
int mc_score_words(mc_model *m, const uint32_t *cps, const uint64_t *off, uint64_t count, uint32_t max_len,
                   uint64_t iter_max, double *out_ml2p, uint64_t *out_split, uint32_t split_words,
                   void *workspace, size_t workspace_bytes, void *stream);
/* Test / measurement hook: which kernel mc_score_words launches.  0 (default): by size -- passwords of at most 32
 * characters with the end symbol's, on a model whose symbols are at most 32 characters, go through
 * k_score_words_fast (characters in LDS, every length of a position looked up in one round of loads from the
 * two-choice symbol table, frames stored only where candidates are left behind), all others through k_score_words
 * (one probe after another, frames in memory).  1: k_score_words always.  Both give the same values, splits and
 * counters (tests/test_gpu_config_shapes.py).  Environment MC_SCORE_WORDS_KERNEL=dfs does the same for a process. */
void mc_score_words_debug_kernel(int which);

/* ---- BPE-PCFG scoring: BpePcfgSim.calc_ml2p over a batch (bpe_simulator.py:118-150).  cls[] holds
 * luds' character class of every code point (0..3 = L U D S), decided on the host with Python's own
 * str.isalpha/isupper/isdigit.  out_ml2p[t] = min over the candidate structures of
 * grammar -log2 p + sum of terminal -log2 p (left to right), 1022.0 when there is none.  (The
 * reference's `break` on a candidate whose grammar value equals 1022.0 depends on set iteration
 * order and is not reproduced; no model of finite probabilities reaches it.) */
int mc_score_bpe(mc_model *m, const uint32_t *cps, const uint8_t *cls, const uint64_t *off, uint64_t count,
                 double *out_ml2p, void *stream);
/* Test / measurement hook, as mc_score_words_debug_kernel: 0 (default) k_score_bpe_fast (passwords of at most 32
 * characters a load round after another, longer ones inside the same launch as k_score_bpe walks them), 1 k_score_bpe
 * always (one load after another).  Same values.  Environment MC_SCORE_BPE_KERNEL=walk does the same for a process. */
void mc_score_bpe_debug_kernel(int which);

/* ---- rank estimation: MonteCarloLib.__init__ (lib4mc/MonteCarloLib.py:10-28): sort the
 * sampled -log2 p ascending, positions = cumsum(2^(key - log2 n_global)).  keys_sorted and
 * positions are outputs [n]; keys_in is left untouched.  n_global = total samples over all
 * GPUs (= n on one GPU): each shard weights with the global n so that per-shard results add.
 *
 * The sort (csrc/mc_sort.hip) counts the copies of heavily repeated values instead of moving them and
 * sends the rest through MSD radix levels; no workgroup of it ever waits for another one, so the build
 * makes progress beside any other work on the GPU (other streams, other processes) and needs no
 * residency guarantee.  The workspace holds two key buffers and the sort's tables (about 2.5 x the keys).
 * Failure is loud: if a table overflows or a count does not match its keys (the caller overwrote keys_in
 * while the build ran), nothing is written outside the buffers, `positions` is filled with NaN and
 * mc_rank_check reports it.
 *
 * The workspace also carries a hint from one build to the next: the list of heavily repeated values the last
 * build of n keys found there is taken again instead of being searched for (keys from one source repeat the same
 * values; any list gives a correct sort, a stale one only a slower one); the search runs again when the tail it
 * leaves has grown by a quarter or is more than three quarters of the keys, every 64th build, and whenever the workspace holds no such list.  Results never
 * depend on it; MC_SORT_REUSE=0 in the environment turns it off.
 */
size_t mc_rank_workspace_bytes(uint64_t n);
int mc_rank_build(const double *keys_in, uint64_t n, uint64_t n_global, double *keys_sorted, double *positions,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Makes a workspace of mc_rank_build or mc_order_keys for n keys forget the list of heavy values it may hold (the
 * next call searches for them): for a caller that hands out workspaces from a pool and wants every call to behave
 * as the first, as the host classes do for the workspaces they allocate themselves.  No counterpart in the reference. */
int mc_workspace_forget(void *workspace, uint64_t n, void *stream);

/* Synchronises `stream` and reads the status of the last mc_rank_build that used `workspace` for n keys:
 * MC_OK, or MC_ERR_WORKSPACE when the build failed on the device (positions are NaN then).  info (may be
 * NULL) receives {error word, heavy values found, tail keys, radix levels used, finish tiles, segments, builds the
 * list of heavy values has served since it was searched for, 0}.
 * No counterpart in the reference (its list.sort() cannot fail). */
int mc_rank_check(const void *workspace, uint64_t n, void *stream, uint64_t info[8]);
/* Test hook: while on, every mc_rank_build takes its device-side failure path (tables "overflow" at once). */
void mc_rank_debug_fail(int on);
/* Test / measurement hook: how mc_rank_build looks for the values that repeat often.  0 (default): by size --
 * from 2^26 keys on a 2^19-key sample is counted and up to 3072 values are taken out of the sort, below that a
 * 16384-key sample and up to 1024 values; 1: the large search wherever the workspace holds its tables (2^22 keys
 * and more); 2: never.  The result is the same sorted array either way. */
void mc_rank_debug_search(int mode);

/* mc_rank_build for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it is the same
 * computation as mc_rank_build (same kernels, same results bit for bit) with the workspace taken from a
 * device array built into the library instead of from the caller; inputs and outputs are still the
 * caller's.  It calls no hipMalloc / hipFree or any allocating routine.  Limits: n <=
 * MC_RANK_NOALLOC_MAX_KEYS (beyond that MC_ERR_ARG with a message; use mc_rank_build); every GPU the
 * library runs on holds MC_RANK_NOALLOC_SCRATCH_BYTES of device memory for the array until the process
 * ends.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race -- serialise them. */
#define MC_RANK_NOALLOC_MAX_KEYS (1ull << 20)
#define MC_RANK_NOALLOC_SCRATCH_BYTES (24ull << 20)
uint64_t mc_rank_noalloc_max_keys(void);
int mc_rank_build_noalloc(const double *keys_in, uint64_t n, uint64_t n_global, double *keys_sorted,
                          double *positions, void *stream);

/* ---- rank estimation from the runs of equal keys ---------------------------------------------------------
 * MonteCarloLib.__init__ + ml2p2rank (lib4mc/MonteCarloLib.py:10-32) again, built from what a query reads.
 * `bisect_right(sorted, x)` (:31) always lands on the END of a run of equal keys, so of the n sorted keys and n
 * positions the reference materialises (:11-27) a query reads one entry per DISTINCT value, and sampled -log2 p
 * repeat heavily (1e7 samples of the headline model hold 170,340 distinct values).  The index is, per distinct
 * value v in ascending order (D of them):
 *     vals[d] = v,   cum_count[d] = #{keys <= v},   cum_pos[d] = sum of 2^(key - log2 n_global) over those keys
 * = positions[cum_count[d] - 1] of mc_rank_build up to the order of the additions (compare at 1e-9 relative;
 * cum_count and vals are exact).  The sorted array is never written: ONE read of the keys (copies of the values
 * that repeat most are counted in LDS, the rest is ordered by value range batch by batch and counted range by
 * range in LDS; csrc/mc_runs.hip), four launches, no workgroup ever waits for another one.
 *   vals / cum_count / cum_pos   outputs (device), `capacity` >= mc_rank_runs_capacity(n) entries each
 *   info                         output (device uint64[8]): {D, error word, keys counted, heavy values, tail keys,
 *                                builds the list of heavy values has served, 0, 0}; the query reads D from it, so
 *                                nothing comes back to the host between build and query
 * Keys: non-negative doubles (as mc_rank_build).  mc_rank_runs_max_keys() = 2^26 is the size up to which the index
 * is the host classes' default.  Beyond it and up to 2^32 - 1 keys mc_rank_runs_build works in two levels (the keys
 * that are no copies of the sample's heavy values go to 64 value-range groups, each group is then built as above,
 * all of them side by side, and the groups' runs are joined: nine launches); correct against mc_rank_build where it
 * was tried (7e7 and 2e8 keys: tests/test_gpu_runs.py, scripts/runs_big_check.py) but not faster below 2e8 keys,
 * and the 1.25e9 samples of the v2 6-gram model are refused (their highest value ranges hold a million keys that
 * are all distinct): nothing takes it by default.  Keys that do not repeat
 * (more than 2048 distinct values in one of the 1024 equal parts of the value range: uniform random numbers) do
 * not suit the index: the build then marks it failed (error word set, D = 0, every query answers NaN, count all ones) and
 * mc_rank_runs_check returns MC_ERR_UNSUITED -- the caller builds with mc_rank_build instead, which sorts anything
 * (montecarlo_amd.engine.DeviceRank does).  The workspace carries the heavy values and the value range of one
 * build to the next build of n keys (as mc_rank_build's does; results never depend on it); mc_rank_runs_forget
 * clears that. */
size_t mc_rank_runs_workspace_bytes(uint64_t n);
uint64_t mc_rank_runs_capacity(uint64_t n);
uint64_t mc_rank_runs_max_keys(void);
int mc_rank_runs_build(const double *keys_in, uint64_t n, uint64_t n_global, double *vals, uint64_t *cum_count,
                       double *cum_pos, uint64_t capacity, uint64_t *info, void *workspace, size_t workspace_bytes,
                       void *stream);
int mc_rank_runs_forget(void *workspace, uint64_t n, void *stream);
/* Synchronises `stream` and copies info to the host (host_info may be NULL): MC_OK, MC_ERR_UNSUITED (see above) or
 * MC_ERR_WORKSPACE (the keys changed while the build ran).  No counterpart in the reference. */
int mc_rank_runs_check(const uint64_t *info, void *stream, uint64_t host_info[8]);
/* ml2p2rank (lib4mc/MonteCarloLib.py:30-32) on the run index: out_cnt = bisect_right(sorted keys, x) =
 * cum_count of the last run at or below x (0: none), out_pos = its cum_pos (0.0: none) -- what mc_rank_query
 * returns on the arrays of mc_rank_build.  A failed index answers NaN with a count of all ones (never 0: a count of 0
 * reads "no sample at or below x", whose estimate is 1 -- a guess number). */
int mc_rank_runs_query(const double *vals, const uint64_t *cum_count, const double *cum_pos, const uint64_t *info,
                       const double *x, uint64_t t, double *out_pos, uint64_t *out_cnt, void *stream);
/* The arrays the reference keeps (lib4mc/MonteCarloLib.py:11, :24-27), written from the runs for callers that
 * look at them (MonteCarloLib sorts the caller's list in place): keys_sorted[n] and / or positions[n] (either
 * may be NULL); positions inside a run of c copies of v grow by 2^(v - log2 n_global) a copy.  A failed index
 * fills both with NaN. */
int mc_rank_runs_expand(const double *vals, const uint64_t *cum_count, const double *cum_pos, const uint64_t *info,
                        uint64_t n, uint64_t n_global, double *keys_sorted, double *positions, void *stream);

/* Development aid, no counterpart in the reference: on != 0 makes the run-index builds that follow add the clock
 * ticks (100 MHz) their workgroups spend per phase to 16 device counters; out (may be NULL) receives the counters
 * gathered so far and clears them.  Synchronises the device. */
int mc_rank_runs_debug_phases(int on, uint64_t out[16]);
/* Development aid: for a build of more than 2^26 keys (two levels: 64 value-range groups), what every group
 * received and made of it: out[9 g] = keys of group g, out[9 g + 1 .. 9 g + 8] = the info words of the group's own
 * index.  `out` holds 64 * 9 words.  Waits for the stream. */
int mc_rank_runs_debug_groups(const void *workspace, uint64_t n, uint64_t *out, void *stream);

/* ml2p2rank (lib4mc/MonteCarloLib.py:30-32) for t queries: cnt = bisect_right(keys_sorted, x),
 * pos = positions[cnt-1] or 0.0 when cnt == 0 (the caller maps a global cnt of 0 to the
 * reference's `1`, after summing shards). */
int mc_rank_query(const double *keys_sorted, const double *positions, uint64_t n, const double *x, uint64_t t,
                  double *out_pos, uint64_t *out_cnt, void *stream);

/* ---- sharded estimation: the one collective of the path -------------------------------------------*/
/* SURVEY.md §8(b)/(e).  The reference is single-process and has no counterpart; what is summed is the
 * `positions[idx - 1]` of ml2p2rank (lib4mc/MonteCarloLib.py:31-32) when every rank holds a shard of the
 * samples and has built its rank arrays with the GLOBAL n (mc_rank_build's n_global).  For callers
 * without torch.distributed: RCCL over xGMI, bound at run time (the library loads without it).
 *   mc_comm_unique_id   rank 0 creates the id and hands its MC_COMM_ID_BYTES bytes to the other ranks by any
 *                       means it has (a file, MPI, a socket)
 *   mc_comm_init        every rank, with its GPU current; blocks until all `world` ranks have called
 *   mc_allreduce_rank   in place on device vectors: pos[t] summed over the ranks, and cnt[t] too unless NULL
 *                       (every weight is positive, so "some rank counted a sample" is "the summed position
 *                       is > 0": mc_guess_numbers takes cnt == NULL); asynchronous on `stream`             */
typedef struct mc_comm mc_comm;
#define MC_COMM_ID_BYTES 128
int mc_comm_unique_id(unsigned char *id);
int mc_comm_init(const unsigned char *id, int rank, int world, mc_comm **out);
int mc_allreduce_rank(mc_comm *c, double *pos, uint64_t *cnt, uint64_t t, void *stream);
void mc_comm_destroy(mc_comm *c);

/* ---- the test set's side of guess-number estimation ----------------------------------------------*/
/* Stable ascending order of t test passwords by -log2 p: what `sorted(..., key=lambda x: x[2])` at the end of
 * MonteCarlo.parse_file (lib4mc/MonteCarloParent.py:68) and `need_resort` of ml2p_iter2gc
 * (lib4mc/MonteCarloLib.py:43-45) compute on the host.  x (device, left untouched); outputs (device):
 * x_sorted[t] and order[t] = index of the i-th password in that order, ties by index (Python's stable sort).
 * Any doubles order as Python compares them (-0.0 == 0.0; a NaN, which no scoring kernel produces, goes to one
 * of the ends).  t < 2^32 - 1.  If the key sort gives up on the device (csrc/mc_sort.hip: a table overflow), every
 * order[i] is 0xFFFFFFFF and every x_sorted[i] NaN: never a wrong order. */
size_t mc_order_workspace_bytes(uint64_t t);
int mc_order_keys(const double *x, uint64_t t, double *x_sorted, uint32_t *order, void *workspace,
                  size_t workspace_bytes, void *stream);
/* The same sort in two halves (lib4mc/MonteCarloParent.py:68 again: mc_order_keys IS the two calls in a row).  The
 * bisection of ml2p_iter2gc needs the scores in ascending order, not which password has which: mc_order_keys_begin
 * runs the key sort and leaves x_values[t] = the scores in ascending order AS NUMBERS (both zeros as +0.0; NaN
 * everywhere if the key sort gave up; may be NULL), so a caller can put its queries (mc_rank_runs_guess_numbers,
 * mc_rank_query) on one stream while mc_order_keys_end finishes the passwords' order on another.  Same x, t, x_sorted
 * and workspace in both; x_sorted holds the sort's own words in between and the caller's bits, as after mc_order_keys,
 * once mc_order_keys_end has run; x_values is a buffer of its own. */
int mc_order_keys_begin(const double *x, uint64_t t, double *x_sorted, double *x_values, void *workspace,
                        size_t workspace_bytes, void *stream);
int mc_order_keys_end(const double *x, uint64_t t, double *x_sorted, uint32_t *order, void *workspace,
                      size_t workspace_bytes, void *stream);
/* The loop of ml2p_iter2gc (lib4mc/MonteCarloLib.py:47-55) over t passwords IN ASCENDING ORDER of -log2 p:
 * estimate_i = pos[i] if cnt[i] > 0 else 1 (cnt == NULL: pos[i] > 0 decides, the sharded form where only the
 * summed positions travel; a NaN position -- a failed index, on any shard -- stays NaN and ends the exact prefix), rank_i = ceil(max(estimate_i, rank_{i-1} + (1 if add1 else 0))), rank_0 = 0.
 * pos / cnt are what mc_rank_query returned (summed over shards).  out_rank[t] (device int64) is exact for the
 * first *out_exact passwords (device uint64, written by the call): the reference computes in Python integers and
 * guess numbers are not bounded by 2^63; estimates ascend, so what does not fit (ceil(estimate) >= 2^62, or a
 * NaN) is a suffix, which the caller finishes from index *out_exact on with the reference's own expression
 * (an unsorted input only makes that suffix start earlier): out_rank[i] for i >= *out_exact holds the fp64 bit
 * pattern of estimate_i, so the one array carries all the caller needs. */
size_t mc_guess_numbers_workspace_bytes(uint64_t t);
int mc_guess_numbers(const double *pos, const uint64_t *cnt, uint64_t t, int add1, int64_t *out_rank,
                     uint64_t *out_exact, void *workspace, size_t workspace_bytes, void *stream);
/* ml2p_iter2gc as the reference writes it -- the bisection inside the loop over the passwords
 * (lib4mc/MonteCarloLib.py:47-55) -- on the run index: what mc_rank_runs_query followed by mc_guess_numbers computes
 * (out_pos, out_cnt, out_rank, out_exact: bit for bit the same), in two launches instead of three; x in ascending
 * order (mc_order_keys), workspace as mc_guess_numbers takes it.  For one GPU: shards sum out_pos before ranking
 * (mc_allreduce_rank) and keep the two calls. */
int mc_rank_runs_guess_numbers(const double *vals, const uint64_t *cum_count, const double *cum_pos, const uint64_t *info,
                               const double *x, uint64_t t, int add1, double *out_pos, uint64_t *out_cnt,
                               int64_t *out_rank, uint64_t *out_exact, void *workspace, size_t workspace_bytes,
                               void *stream);

/* ---- threshold enumeration ------------------------------------------------------------------*/
/* Replaces the recursive enumerators (nwords/nwords_enumerator.py:15-42 `enumerator`/`iterate` for a
 * MC_KIND_NGRAM model, backwords/backwords_enumerator.py:15-45 for MC_KIND_BACKOFF): every string whose
 * -log2 p (summed in path order, fp64) is < threshold, closed by the end symbol at cur_len >= min_len, in
 * the order the reference's depth-first loop writes them.  A node with cur_len > max_len is not expanded
 * (30 and 256 in the two references).  Fixed order: an end symbol before min_len is dropped; backoff: it is
 * appended and the walk continues from the context the table gives (the reference's behaviour, kept).
 * Runs as a level-by-level frontier expansion on the model's GPU; the levels are library-owned device
 * buffers, at most `max_nodes` nodes in total (MC_ERR_NOMEM beyond).  Synchronises `stream`. */
typedef struct mc_enum_result mc_enum_result;
enum {
    MC_ENUM_ML2P = 0, /* double[n_out]            -log2 p of every written string                  */
    MC_ENUM_LEN = 1,  /* uint32[n_out]            its length in symbols                            */
    MC_ENUM_SYMS = 2  /* uint32[n_out * stride]   its symbol ids (row-major, `stride` per string)  */
};
int mc_enumerate(const mc_model *m, double threshold, uint32_t min_len, uint32_t max_len, uint64_t max_nodes,
                 void *stream, mc_enum_result **out);
/* info[0..3] = strings written, stride of MC_ENUM_SYMS, nodes expanded, levels */
int mc_enum_info(const mc_enum_result *r, uint64_t info[4]);
/* copies up to `cap` bytes to HOST memory `dst`, returns the full size in bytes */
int64_t mc_enum_fetch(const mc_enum_result *r, int what, void *dst, size_t cap);
void mc_enum_free(mc_enum_result *r);

#ifdef __cplusplus
}
#endif
#endif /* MI355MC_H */
