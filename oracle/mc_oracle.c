/*
 * mc_oracle.c -- CPU restatement (plain C) of the reference's hot path on flattened tables.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under montecarlo_amd/ links or loads it.
 *
 * It follows the REFERENCE's algorithm step by step, not the GPU design: the context of
 * every step is found by looking suffixes of the whole history up in a context map
 * (backwords_simulator.py:28-41 / nwords_simulator.py:30-34), the draw is a plain
 * bisect_right over the row's cumulative sums (lib4mc/ProbLib.py:40-41), scoring walks
 * suffixes from the longest (backwords_simulator.py:31-38).  There are no next-state
 * links, guide tables or transition hashes here, so agreement with the HIP path checks
 * those precomputations.  Uniforms: Philox4x32-10 as defined in oracle/philox.py.
 *
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the golden vectors
 * recorded from the reference (the JSON files under tests/golden).
 *
 * Input layout = struct mc_model_desc of include/mi355mc.h (the flattened dict-of-dict).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mi355mc.h"

typedef struct orc_model {
    mc_model_desc d;
    uint32_t *sorted; /* row ids sorted by context (length, then lexicographic) */
    uint32_t maxctx;  /* longest context: longer suffixes of a history can never be rows */
} orc_model;

static const mc_model_desc *g_cmp_desc;

static int ctx_cmp_rows(const void *pa, const void *pb) {
    const mc_model_desc *d = g_cmp_desc;
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    uint32_t la = d->ctx_off[a + 1] - d->ctx_off[a], lb = d->ctx_off[b + 1] - d->ctx_off[b];
    if (la != lb) return la < lb ? -1 : 1;
    const uint32_t *ua = d->ctx_units + d->ctx_off[a], *ub = d->ctx_units + d->ctx_off[b];
    for (uint32_t i = 0; i < la; ++i)
        if (ua[i] != ub[i]) return ua[i] < ub[i] ? -1 : 1;
    return 0;
}

/* `ctx in self.nwords`: binary search over the sorted contexts; MC_NONE when absent */
static uint32_t ctx_find(const orc_model *m, const uint32_t *u, uint32_t n) {
    const mc_model_desc *d = &m->d;
    int64_t lo = 0, hi = (int64_t)d->n_ctx - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        uint32_t r = m->sorted[mid];
        uint32_t lr = d->ctx_off[r + 1] - d->ctx_off[r];
        int c = 0;
        if (lr != n) c = lr < n ? -1 : 1;
        else {
            const uint32_t *ur = d->ctx_units + d->ctx_off[r];
            for (uint32_t i = 0; i < n && !c; ++i)
                if (ur[i] != u[i]) c = ur[i] < u[i] ? -1 : 1;
        }
        if (c == 0) return r;
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return MC_NONE;
}

orc_model *orc_model_new(const mc_model_desc *d) {
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    m->d = *d; /* arrays stay owned by the caller */
    m->sorted = (uint32_t *)malloc(sizeof(uint32_t) * d->n_ctx);
    for (uint32_t r = 0; r < d->n_ctx; ++r) {
        m->sorted[r] = r;
        uint32_t l = d->ctx_off[r + 1] - d->ctx_off[r];
        if (l > m->maxctx) m->maxctx = l;
    }
    g_cmp_desc = d;
    qsort(m->sorted, d->n_ctx, sizeof(uint32_t), ctx_cmp_rows);
    return m;
}

void orc_model_free(orc_model *m) {
    if (m) {
        free(m->sorted);
        free(m);
    }
}

/* does row r list symbol s?  returns the entry index or -1 (`left in self.nwords.get(prev)[0]`) */
static int64_t row_find(const mc_model_desc *d, uint32_t r, uint32_t s) {
    for (uint64_t e = d->row_off[r]; e < d->row_off[r + 1]; ++e)
        if (d->ent_sym[e] == s) return (int64_t)e;
    return -1;
}

/* _get_prefix: the row that applies to history h[0..n), optionally requiring transition s */
static uint32_t context_row(const orc_model *m, const uint32_t *h, uint32_t n, int need_sym, uint32_t s) {
    const mc_model_desc *d = &m->d;
    if (d->kind == MC_KIND_BACKOFF) {
        /* longest suffix first (backwords_simulator.py:31-38); suffixes longer than the longest
         * context in the table cannot be rows, so the scan starts there */
        for (uint32_t i = n > m->maxctx ? n - m->maxctx : 0; i < n; ++i) {
            uint32_t r = ctx_find(m, h + i, n - i);
            if (r == MC_NONE) continue;
            if (need_sym && row_find(d, r, s) < 0) continue;
            return r;
        }
        return ctx_find(m, h, 0); /* tar = tuple() (:39-40) */
    }
    uint32_t keep = n < d->order ? n : d->order; /* nwords_simulator.py:30-34 / v2 :65-68 */
    return ctx_find(m, h + (n - keep), keep);
}

/* ---- Philox4x32-10 (oracle/philox.py) ---------------------------------------------------- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        if (r) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0;
        c[1] = n1;
        c[2] = n2;
        c[3] = n3;
    }
}

double orc_uniform(uint64_t seed, uint64_t id, uint32_t draw) {
    uint32_t c[4] = {(uint32_t)id, (uint32_t)(id >> 32), draw >> 1, 0};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t a = (draw & 1) ? c[2] : c[0], b = (draw & 1) ? c[3] : c[1];
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) / 9007199254740992.0;
}

/* ---- sample1 loop (nwords_simulator.py:92-116, nwords_simulator_v2.py:60-82) ---------------
 * uniforms == NULL: Philox(seed, first_id + i, k); else uniforms[i*max_draws + k].
 * A sample that exhausts max_draws or meets a missing row yields NaN. */
int orc_sample(const orc_model *m, uint64_t first_id, uint64_t count, uint64_t seed, uint32_t max_draws,
               const double *uniforms, double *out_ml2p, uint32_t *out_len, uint32_t *out_draws, uint32_t *out_syms,
               uint32_t stride) {
    const mc_model_desc *d = &m->d;
    uint32_t cap = d->n_start + m->maxctx + 4096;
    uint32_t *units = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap); /* history, in context units */
    const int keep_len = (d->flags & MC_FLAG_KEEP_LEN_ON_RESTART) != 0;
    for (uint64_t i = 0; i < count; ++i) {
        uint32_t n = d->n_start, plen = 0, nsym = 0, draw = 0;
        double acc = 0.0;
        int ok = 0;
        memcpy(units, d->start_units, sizeof(uint32_t) * d->n_start);
        for (;;) {
            if (draw >= max_draws) break;
            uint32_t row = context_row(m, units, n, 0, 0);
            if (row == MC_NONE) break;
            double r = uniforms ? uniforms[i * (uint64_t)max_draws + draw] : orc_uniform(seed, first_id + i, draw);
            ++draw;
            const double *cum = d->ent_cum + d->row_off[row];
            uint64_t L = d->row_off[row + 1] - d->row_off[row];
            double total = cum[L - 1];
            double u = 0.0 + (total - 0.0) * r; /* random.uniform(0, total) */
            uint64_t lo = 0, hi = L;             /* bisect_right */
            while (lo < hi) {
                uint64_t mid = (lo + hi) / 2;
                if (u < cum[mid]) hi = mid; else lo = mid + 1;
            }
            if (lo >= L) lo = L - 1; /* reference: IndexError; build: clamp */
            uint64_t e = d->row_off[row] + lo;
            uint32_t s = d->ent_sym[e];
            acc += d->ent_ml2p[e];
            int restart = 0;
            if (s == d->end_sym) {
                if (plen >= d->min_len) {
                    ok = 1;
                    break;
                }
                restart = 1;
            } else {
                if (out_syms && nsym < stride) out_syms[i * (uint64_t)stride + nsym] = s;
                ++nsym;
                uint32_t q0 = d->sym_unit_off ? d->sym_unit_off[s] : 0, q1 = d->sym_unit_off ? d->sym_unit_off[s + 1] : 1;
                for (uint32_t q = q0; q < q1; ++q) {
                    if (n == cap) { /* keep only what any context can still see */
                        memmove(units, units + (n - m->maxctx), sizeof(uint32_t) * m->maxctx);
                        n = m->maxctx;
                    }
                    units[n++] = d->sym_unit_off ? d->sym_units[q] : s;
                }
                plen += d->sym_chars ? d->sym_chars[s] : 1;
                if (plen >= d->max_len) restart = 1;
            }
            if (restart) {
                n = d->n_start;
                memcpy(units, d->start_units, sizeof(uint32_t) * d->n_start);
                acc = 0.0;
                nsym = 0;
                if (!keep_len) plen = 0;
            }
        }
        out_ml2p[i] = ok ? acc : NAN;
        if (out_len) out_len[i] = nsym;
        if (out_draws) out_draws[i] = draw;
    }
    free(units);
    return 0;
}

/* ---- calc_ml2p for single-character alphabets: one parse (nwords_simulator.py:36-90) -------- */
int orc_score(const orc_model *m, const uint32_t *syms, const uint64_t *off, uint64_t count, double *out) {
    const mc_model_desc *d = &m->d;
    for (uint64_t t = 0; t < count; ++t) {
        uint64_t L = off[t + 1] - off[t];
        uint32_t *h = (uint32_t *)malloc(sizeof(uint32_t) * (d->n_start + L + 2));
        uint32_t n = d->n_start;
        memcpy(h, d->start_units, sizeof(uint32_t) * d->n_start);
        double acc = 0.0;
        int ok = 1;
        for (uint64_t j = 0; j <= L && ok; ++j) {
            uint32_t s = j < L ? syms[off[t] + j] : d->end_sym;
            if (s >= d->n_sym) {
                ok = 0;
                break;
            }
            uint32_t row = context_row(m, h, n, 1, s);
            int64_t e = row == MC_NONE ? -1 : row_find(d, row, s);
            if (e < 0) {
                ok = 0;
                break;
            }
            acc += d->ent_ml2p[e];
            h[n++] = s;
        }
        out[t] = (ok && 1022.0 > acc) ? acc : 1022.0;
        free(h);
    }
    return 0;
}

/* ---- MonteCarloLib (lib4mc/MonteCarloLib.py:10-32) ------------------------------------------- */
static int dbl_cmp(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

void orc_rank_build(const double *keys_in, uint64_t n, uint64_t n_global, double *keys_sorted, double *positions) {
    memcpy(keys_sorted, keys_in, sizeof(double) * n);
    qsort(keys_sorted, n, sizeof(double), dbl_cmp);
    double logn = log2((double)n_global), run = 0.0;
    for (uint64_t i = 0; i < n; ++i) { /* sequential cumsum, smallest terms first */
        run += pow(2.0, keys_sorted[i] - logn);
        positions[i] = run;
    }
}

void orc_rank_query(const double *keys_sorted, const double *positions, uint64_t n, const double *x, uint64_t t,
                    double *out_pos, uint64_t *out_cnt) {
    for (uint64_t q = 0; q < t; ++q) {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (x[q] < keys_sorted[mid]) hi = mid; else lo = mid + 1;
        }
        out_cnt[q] = lo;
        out_pos[q] = lo ? positions[lo - 1] : 0.0;
    }
}
