// Table compiler: dict-of-dict model (flattened by the caller) -> walk automaton.
//
// The reference recomputes, for every emitted symbol, which table row applies
// (`_get_prefix`: nwords_simulator.py:30-34, backwords_simulator.py:28-41 -- tuple
// builds + dict probes over every suffix of the history) and bisects the row's
// cumulative sums (lib4mc/ProbLib.py:40-41).  On the GPU both are replaced by data:
//   * next[e]   : the row that applies after taking entry e (one u32 per entry);
//   * parent[r] : the next shorter row that is a suffix of row r's context (backoff);
//   * draw      : per row, 2^gbits buckets over the uniform r; a bucket is one 16-byte record
//                 that either IS the entry every r of the bucket lands on (ml2p, next row's
//                 buckets, symbol) or brackets the few candidates, which are then told apart
//                 by thresholds precomputed in r's own domain (include/mi355mc.h, mc_draw_rec).
//                 The result is the index bisect_right(cum, total*r) returns, bit for bit.
//   * trans     : open-addressed (row,symbol) -> entry hash for scoring;
//   * ctxhash   : open-addressed 64-bit context hash -> row index (by-context lookups).
// Host-only code: no HIP here, so it is testable in a container without a GPU.
#include "mc_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <string>

static thread_local std::string g_err;

void mc_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char *mc_last_error(void) { return g_err.c_str(); }
extern "C" int mc_abi_version(void) { return MC_ABI_VERSION; }

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" uint64_t mc_ctx_hash(const uint32_t *units, uint32_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n << 32);
    for (uint32_t i = 0; i < n; ++i) h = mix64(h ^ ((uint64_t)units[i] + 0x632BE59BD9B4E019ull));
    if (h == ~0ull) h = ~0ull - 1;  // ~0 marks an empty slot
    return h;
}

uint64_t mc_trans_hash(uint64_t key) { return mix64(key + 0x9E3779B97F4A7C15ull); }

// FNV-1a over code points, one step per character: the scoring kernel extends it character by
// character as it tries longer and longer splits.
extern "C" uint64_t mc_text_hash(const uint32_t *cps, uint32_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (uint32_t i = 0; i < n; ++i) h = (h ^ cps[i]) * 0x100000001B3ull;
    if (h == ~0ull) h = ~0ull - 1;
    return h;
}

namespace {

uint64_t pow2ceil(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Smallest double r in [0, 1] with fl(total * r) >= c (c <= total).  fl(total * r) is monotone
// in r and non-negative doubles order like their bit patterns, so bisect on the bits.
double draw_threshold(double total, double c) {
    uint64_t lo = 0, hi;
    const double one = 1.0;
    std::memcpy(&hi, &one, sizeof hi);
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        double r;
        std::memcpy(&r, &mid, sizeof r);
        volatile double prod = total * r;  // one rounding to double, as the reference's total * random()
        if (prod >= c) hi = mid; else lo = mid + 1;
    }
    double r;
    std::memcpy(&r, &lo, sizeof r);
    return r;
}

struct CtxIndex {
    const mc_model_desc *d;
    std::vector<mc_slot> slots;
    uint64_t mask;

    bool same(uint32_t row, const uint32_t *u, uint32_t n) const {
        uint32_t a = d->ctx_off[row], b = d->ctx_off[row + 1];
        return b - a == n && (n == 0 || std::memcmp(d->ctx_units + a, u, n * sizeof(uint32_t)) == 0);
    }
    // returns MC_NONE when absent
    uint32_t find(const uint32_t *u, uint32_t n) const {
        uint64_t h = mc_ctx_hash(u, n);
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            const mc_slot &s = slots[i];
            if (s.key == ~0ull) return MC_NONE;
            if (s.key == h && same(s.val, u, n)) return s.val;
        }
    }
    int build(const mc_model_desc *desc) {
        d = desc;
        uint64_t cap = pow2ceil(std::max<uint64_t>(16, 2ull * desc->n_ctx));
        mask = cap - 1;
        slots.assign(cap, mc_slot{~0ull, 0, 0});
        for (uint32_t r = 0; r < desc->n_ctx; ++r) {
            uint32_t a = desc->ctx_off[r], n = desc->ctx_off[r + 1] - a;
            uint64_t h = mc_ctx_hash(desc->ctx_units + a, n);
            for (uint64_t i = h & mask;; i = (i + 1) & mask) {
                mc_slot &s = slots[i];
                if (s.key == ~0ull) {
                    s = mc_slot{h, r, n};
                    break;
                }
                if (s.key == h) {
                    if (same(s.val, desc->ctx_units + a, n)) {
                        mc_set_error("context rows %u and %u are the same context", s.val, r);
                    } else {
                        mc_set_error("64-bit context hash collision between rows %u and %u", s.val, r);
                    }
                    return MC_ERR_TABLE;
                }
            }
        }
        return MC_OK;
    }
};

}  // namespace

extern "C" int mc_table_compile(const mc_model_desc *d, mc_table **out) {
    if (!d || !out) {
        mc_set_error("mc_table_compile: null argument");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    if (d->kind != MC_KIND_NGRAM && d->kind != MC_KIND_BACKOFF && d->kind != MC_KIND_V2 && d->kind != MC_KIND_BPE) {
        mc_set_error("unknown model kind %u", d->kind);
        return MC_ERR_ARG;
    }
    if (!d->ctx_off || !d->row_off || !d->ent_sym || !d->ent_cum || !d->ent_ml2p || (d->n_start && !d->start_units) ||
        (d->ctx_off[d->n_ctx] && !d->ctx_units)) {
        mc_set_error("mc_table_compile: null array in descriptor");
        return MC_ERR_ARG;
    }
    if (d->n_ctx == 0 || d->n_ctx >= MC_NONE || d->n_ent == 0 || d->n_ent >= 0x7FFFFFFFull || d->n_sym == 0 ||
        d->n_sym >= 0x7FFFFFFFu || d->end_sym >= d->n_sym) {
        mc_set_error("mc_table_compile: sizes out of range (n_ctx=%u n_ent=%llu n_sym=%u end_sym=%u)", d->n_ctx,
                     (unsigned long long)d->n_ent, d->n_sym, d->end_sym);
        return MC_ERR_ARG;
    }
    if (d->kind == MC_KIND_BPE) {
        const uint64_t n_struct = d->row_off[1];
        if (!d->seg_off || !d->seg_row || !d->seg_len || !d->sym_text_off || (d->n_labels && (!d->lab_off ||
            !d->lab_units || !d->lab_cand_off || !d->lab_cand)) || (d->n_np && (!d->np_len || !d->np_off || !d->np_cand))) {
            mc_set_error("MC_KIND_BPE needs seg_*, sym_text_* and the label / not_parsed arrays");
            return MC_ERR_ARG;
        }
        for (uint64_t i = 0; i < n_struct; ++i)
            for (uint32_t k = d->seg_off[i]; k < d->seg_off[i + 1]; ++k)
                if (d->seg_row[k] != MC_NONE && (d->seg_row[k] == 0 || d->seg_row[k] >= d->n_ctx)) {
                    mc_set_error("structure %llu: segment row %u out of range", (unsigned long long)i, d->seg_row[k]);
                    return MC_ERR_ARG;
                }
        for (uint32_t j = 0; j < d->n_labels; ++j)
            for (uint32_t k = d->lab_cand_off[j]; k < d->lab_cand_off[j + 1]; ++k)
                if (d->lab_cand[k] >= n_struct) {
                    mc_set_error("label %u: candidate %u is not a structure", j, d->lab_cand[k]);
                    return MC_ERR_ARG;
                }
        for (uint32_t j = 0; j < d->n_np; ++j)
            for (uint32_t k = d->np_off[j]; k < d->np_off[j + 1]; ++k)
                if (d->np_cand[k] >= n_struct) {
                    mc_set_error("not_parsed %u: candidate %u is not a structure", j, d->np_cand[k]);
                    return MC_ERR_ARG;
                }
    }
    if (d->kind != MC_KIND_BACKOFF && d->kind != MC_KIND_BPE && d->order == 0) {
        mc_set_error("fixed-order models need order >= 1 (n >= 2)");
        return MC_ERR_ARG;
    }
    if (d->row_off[0] != 0 || d->row_off[d->n_ctx] != d->n_ent || d->ctx_off[0] != 0) {
        mc_set_error("row_off/ctx_off must start at 0 and row_off must end at n_ent");
        return MC_ERR_ARG;
    }
    for (uint32_t r = 0; r < d->n_ctx; ++r) {
        if (d->row_off[r + 1] <= d->row_off[r]) {
            mc_set_error("row %u is empty (expand_2d drops empty rows, ProbLib.py:14-15)", r);
            return MC_ERR_TABLE;
        }
        if (d->ctx_off[r + 1] < d->ctx_off[r]) {
            mc_set_error("ctx_off not monotone at row %u", r);
            return MC_ERR_ARG;
        }
    }
    for (uint64_t e = 0; e < d->n_ent; ++e)
        if (d->ent_sym[e] >= d->n_sym) {
            mc_set_error("entry %llu has symbol id %u >= n_sym", (unsigned long long)e, d->ent_sym[e]);
            return MC_ERR_ARG;
        }

    CtxIndex idx;
    int rc = idx.build(d);
    if (rc != MC_OK) return rc;

    mc_table *t = new mc_table();
    t->kind = d->kind;
    t->order = d->order;
    t->n_ctx = d->n_ctx;
    t->n_sym = d->n_sym;
    t->n_ent = d->n_ent;
    t->end_sym = d->end_sym;
    t->min_len = d->min_len;
    t->max_len = d->max_len;
    t->flags = d->flags;
    t->ctxhash = idx.slots;
    t->sym_chars.assign(d->n_sym, 1);
    t->multichar = false;
    t->len_in_rec = false;
    uint32_t longest_sym = 1;
    if (d->sym_chars)
        for (uint32_t s = 0; s < d->n_sym; ++s) {
            t->sym_chars[s] = d->sym_chars[s];
            if (d->sym_chars[s] != 1 && s != d->end_sym) t->multichar = true;
            if (s != d->end_sym) longest_sym = std::max(longest_sym, d->sym_chars[s]);
        }

    const bool backoff = d->kind == MC_KIND_BACKOFF;
    const bool bpe = d->kind == MC_KIND_BPE;
    // the row that applies to a history `u[0..n)`, by the model's own context rule
    auto state_of = [&](const uint32_t *u, uint32_t n) -> uint32_t {
        if (bpe) return n == 1 ? idx.find(u, 1) : MC_NONE;
        if (backoff) {
            for (uint32_t i = 0; i <= n; ++i) {  // longest suffix first, the empty context last
                uint32_t r = idx.find(u + i, n - i);
                if (r != MC_NONE) return r;
            }
            return MC_NONE;
        }
        uint32_t keep = std::min(n, d->order);
        return idx.find(u + (n - keep), keep);
    };

    t->root = idx.find(nullptr, 0);
    t->start_state = state_of(d->start_units, d->n_start);

    if (backoff) {
        // Exactness of the automaton: state(h+s) must equal the longest suffix of state(h)+s that
        // is a row.  That holds iff the rows are prefix-closed (w+s a row => w a row), which every
        // trainer of the reference guarantees (count[w][s] >= total(w+s) >= threshold).
        for (uint32_t r = 0; r < d->n_ctx; ++r) {
            uint32_t a = d->ctx_off[r], n = d->ctx_off[r + 1] - a;
            if (n == 0) continue;
            if (idx.find(d->ctx_units + a, n - 1) == MC_NONE) {
                mc_set_error("backoff rows are not prefix-closed: row %u (context length %u) has no row for its "
                             "context minus the last symbol; the compiled walk would not be exact",
                             r, n);
                delete t;
                return MC_ERR_TABLE;
            }
        }
    }

    // ---- state records, payloads with next-state links ---------------------------------------
    if (d->n_sym > MC_MAX_SYMBOLS) {
        mc_set_error("%u symbols: the draw records hold 24-bit symbol ids (at most %u)", d->n_sym, MC_MAX_SYMBOLS);
        delete t;
        return MC_ERR_TABLE;
    }
    uint32_t gf = d->guide_factor ? d->guide_factor : 2;
    t->state.resize(d->n_ctx);
    t->cum.assign(d->ent_cum, d->ent_cum + d->n_ent);
    t->payload.resize(d->n_ent);
    t->max_row = 0;
    std::vector<uint32_t> seq;
    uint64_t n_buckets = 0;
    for (uint32_t r = 0; r < d->n_ctx; ++r) {
        const uint64_t e0 = d->row_off[r], e1 = d->row_off[r + 1];
        const uint32_t L = (uint32_t)(e1 - e0);
        const uint32_t ca = d->ctx_off[r], cn = d->ctx_off[r + 1] - ca;
        const double *cum = d->ent_cum + e0;
        mc_state_rec &st = t->state[r];
        st.total = cum[L - 1];
        st.ent_off = (uint32_t)e0;
        st.len = L;
        st.ctx_len = cn;
        st.parent = MC_NONE;
        t->max_row = std::max(t->max_row, L);
        for (uint32_t i = 1; i < L; ++i)
            if (!(cum[i] >= cum[i - 1])) {
                mc_set_error("row %u: cumulative sums decrease at entry %u", r, i);
                delete t;
                return MC_ERR_TABLE;
            }
        if (!(st.total > 0.0) || !(st.total < 1e300)) {
            mc_set_error("row %u: total probability mass is not a positive finite number", r);
            delete t;
            return MC_ERR_TABLE;
        }
        if (backoff && cn > 0) {
            for (uint32_t i = 1; i <= cn; ++i) {
                uint32_t p = idx.find(d->ctx_units + ca + i, cn - i);
                if (p != MC_NONE) {
                    st.parent = p;
                    break;
                }
            }
        }
        st.gbits = 0;      // decided below from the row's expected visits
        st.guide_off = 0;  // assigned below, hottest rows first
        for (uint64_t e = e0; e < e1; ++e) {
            uint32_t s = d->ent_sym[e];
            seq.assign(d->ctx_units + ca, d->ctx_units + ca + cn);
            if (d->sym_unit_off) {
                for (uint32_t q = d->sym_unit_off[s]; q < d->sym_unit_off[s + 1]; ++q) seq.push_back(d->sym_units[q]);
            } else {
                seq.push_back(s);
            }
            mc_payload &p = t->payload[e];
            p.ml2p = d->ent_ml2p[e];
            p.next = bpe ? MC_NONE : state_of(seq.data(), (uint32_t)seq.size());
            p.sym = s | (s == d->end_sym ? 0x80000000u : 0u);
        }
    }
    // ---- expected visits of every row per sampled string: propagated from the start row along the
    // next-state links (restarts ignored).  An estimate that only steers speed, never results: it
    // decides how many buckets a row gets and where they are placed.
    std::vector<double> visits(d->n_ctx, 0.0);
    {
        std::vector<double> cur(d->n_ctx, 0.0), nxt(d->n_ctx, 0.0);
        if (t->start_state != MC_NONE && !bpe) {
            cur[t->start_state] = 1.0;
            for (int step = 0; step < 48; ++step) {
                std::fill(nxt.begin(), nxt.end(), 0.0);
                double alive = 0.0;
                for (uint32_t r = 0; r < d->n_ctx; ++r) {
                    const double v = cur[r];
                    if (v <= 0.0) continue;
                    visits[r] += v;
                    const mc_state_rec &st = t->state[r];
                    double prev = 0.0;
                    for (uint32_t j = 0; j < st.len; ++j) {
                        const double c = t->cum[st.ent_off + j];
                        const mc_payload &p = t->payload[st.ent_off + j];
                        if (!(p.sym & 0x80000000u) && p.next != MC_NONE) {
                            const double w = v * (c - prev) / st.total;
                            nxt[p.next] += w;
                            alive += w;
                        }
                        prev = c;
                    }
                }
                cur.swap(nxt);
                if (alive < 1e-9) break;
            }
        } else if (bpe) {
            // the grammar row once per string; a terminal row once per segment that names it
            visits[0] = 1.0;
            const mc_state_rec &g = t->state[0];
            double prev = 0.0;
            for (uint32_t j = 0; j < g.len; ++j) {
                const double pj = (t->cum[j] - prev) / g.total;
                prev = t->cum[j];
                for (uint32_t k = d->seg_off[j]; k < d->seg_off[j + 1]; ++k)
                    if (d->seg_row[k] != MC_NONE) visits[d->seg_row[k]] += pj;
            }
        }
    }
    // ---- buckets per row.  A bucket that straddles a threshold costs the sampling kernel a second
    // dependent gather, and a row of L entries has at most L-1 such buckets however many it owns: the
    // share of a row's draws that pay is ~ (L-1) / buckets.  Minimising sum(visits * (L-1) / buckets)
    // under a budget of guide_factor * n_ent buckets gives buckets ~ sqrt(visits * (L-1)); every row
    // keeps at least L buckets (rounded up to 2^k).
    {
        const double budget = (double)gf * (double)d->n_ent;
        auto total_for = [&](double K) {
            double tot = 0.0;
            for (uint32_t r = 0; r < d->n_ctx; ++r) {
                const uint32_t L = t->state[r].len;
                if (L <= 1) { tot += 1.0; continue; }
                double want = std::max((double)L, K * std::sqrt(visits[r] * (double)(L - 1)));
                tot += (double)pow2ceil((uint64_t)std::min(want, 1048576.0));
            }
            return tot;
        };
        double lo = 0.0, hi = 1.0;
        while (total_for(hi) < budget && hi < 1e12) hi *= 2.0;
        for (int it = 0; it < 50; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (total_for(mid) <= budget) lo = mid; else hi = mid;
        }
        for (uint32_t r = 0; r < d->n_ctx; ++r) {
            const uint32_t L = t->state[r].len;
            uint32_t gbits = 0;
            if (L > 1) {
                double want = std::max((double)L, lo * std::sqrt(visits[r] * (double)(L - 1)));
                const uint64_t K = pow2ceil((uint64_t)std::min(want, 1048576.0));
                while ((1ull << gbits) < K) ++gbits;
                if (gbits > 20) gbits = 20;  // (k_sample takes a draw's bucket from the first Philox word of its uniform: <= 27)
            }
            t->state[r].gbits = gbits;
            n_buckets += 1ull << gbits;
        }
        if (n_buckets >= 0xFFFFFFF0ull) {
            mc_set_error("draw table exceeds 2^32 buckets; lower guide_factor");
            delete t;
            return MC_ERR_TABLE;
        }
    }
    // ---- bucket placement: rows in descending order of expected visits per bucket, so that the
    // head of the draw table is what the sampling kernel keeps in LDS.
    {
        std::vector<uint32_t> order(d->n_ctx);
        for (uint32_t r = 0; r < d->n_ctx; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return visits[a] / (double)(1u << t->state[a].gbits) > visits[b] / (double)(1u << t->state[b].gbits);
        });
        uint64_t at = 0;
        for (uint32_t r : order) {
            t->state[r].guide_off = (uint32_t)at;
            at += 1ull << t->state[r].gbits;
        }
    }
    if (t->start_state != MC_NONE) {
        t->start_draw = t->state[t->start_state].guide_off;
        t->start_gbits = t->state[t->start_state].gbits;
    } else {
        t->start_draw = MC_NONE;
        t->start_gbits = 0;
    }

    // a symbol's length beside the next row (MC_DRAW_LEN_SHIFT): the sampler's length count then costs no read
    t->len_in_rec = t->multichar && !bpe && n_buckets < MC_DRAW_OFF_MASK && longest_sym <= MC_DRAW_LEN_MAX &&
                    !getenv("MC_NO_LEN_IN_REC");
    // ---- draw table: thresholds in the uniform's domain, per-entry records, buckets ----------
    t->rcum.resize(d->n_ent);
    t->spay.resize(d->n_ent);
    t->draw.resize(n_buckets);
    t->n_direct = t->n_one = t->n_multi = 0;
    for (uint32_t r = 0; r < d->n_ctx; ++r) {
        const mc_state_rec &st = t->state[r];
        const uint64_t e0 = st.ent_off;
        const uint32_t L = st.len;
        double *T = t->rcum.data() + e0;
        for (uint32_t j = 0; j + 1 < L; ++j) T[j] = draw_threshold(st.total, t->cum[e0 + j]);
        T[L - 1] = 2.0;  // never reached: u == cum[-1] is clamped to the last entry
        for (uint32_t j = 0; j < L; ++j) {
            const mc_payload &p = t->payload[e0 + j];
            mc_draw_rec &q = t->spay[e0 + j];
            std::memcpy(&q.w[0], &p.ml2p, sizeof(double));
            uint32_t nb = 0;
            q.w[2] = MC_NONE;
            if (bpe) {
                q.w[2] = (uint32_t)(e0 + j);  // which row comes next depends on the structure: keep the entry
            } else if (p.next != MC_NONE) {
                q.w[2] = t->state[p.next].guide_off;
                nb = t->state[p.next].gbits;
            }
            if (t->len_in_rec && !(p.sym & MC_DRAW_END)) {
                const uint32_t off = q.w[2] == MC_NONE ? MC_DRAW_OFF_MASK : q.w[2];
                q.w[2] = (t->sym_chars[p.sym & 0x00FFFFFFu] << MC_DRAW_LEN_SHIFT) | off;
            }
            q.w[3] = (p.sym & MC_DRAW_END) | ((uint32_t)MC_DRAW_DIRECT << 29) | (nb << 24) | (p.sym & 0x00FFFFFFu);
        }
        const uint32_t K = 1u << st.gbits;
        for (uint32_t k = 0; k < K; ++k) {
            const double a = (double)k / (double)K, b = (double)(k + 1) / (double)K;  // exact
            const uint32_t lo = (uint32_t)(std::upper_bound(T, T + (L - 1), a) - T);
            const uint32_t hi = (uint32_t)(std::lower_bound(T, T + (L - 1), b) - T);
            mc_draw_rec &q = t->draw[(uint64_t)st.guide_off + k];
            if (hi <= lo) {
                q = t->spay[e0 + lo];
                ++t->n_direct;
            } else if (hi == lo + 1) {
                std::memcpy(&q.w[0], &T[lo], sizeof(double));
                q.w[2] = (uint32_t)(e0 + lo);
                q.w[3] = (uint32_t)MC_DRAW_ONE << 29;
                ++t->n_one;
            } else {
                q.w[0] = hi - lo;
                q.w[1] = 0;
                q.w[2] = (uint32_t)(e0 + lo);
                q.w[3] = (uint32_t)MC_DRAW_MULTI << 29;
                ++t->n_multi;
            }
        }
    }

    // ---- symbol text -> symbol id (the `left in self.words` lookup of _structures) ----------
    t->max_word_len = 0;
    // the start symbol as a transition id (contexts and transitions share ids except for V2)
    t->start_sym = (d->kind != MC_KIND_V2 && d->n_start) ? d->start_units[0] : MC_NONE;
    if (d->sym_text_off) {
        t->sym_text_off.assign(d->sym_text_off, d->sym_text_off + d->n_sym + 1);
        t->sym_text.assign(d->sym_text, d->sym_text + d->sym_text_off[d->n_sym]);
        // (a quarter full at most: a look-up of a text that is no symbol -- most of them -- ends at its first or second slot)
        uint64_t wcap = pow2ceil(std::max<uint64_t>(16, 4ull * d->n_sym));
        t->wordhash.assign(wcap, mc_slot{~0ull, 0, 0});
        for (uint32_t s = 0; s < d->n_sym; ++s) {
            const uint32_t a = d->sym_text_off[s], n = d->sym_text_off[s + 1] - a;
            if (n == 0) continue;  // the empty string is never a split (index starts at 1)
            t->max_word_len = std::max(t->max_word_len, n);
            const uint64_t h = mc_text_hash(d->sym_text + a, n);
            for (uint64_t i = h & (wcap - 1);; i = (i + 1) & (wcap - 1)) {
                mc_slot &sl = t->wordhash[i];
                if (sl.key == ~0ull) {
                    sl = mc_slot{h, s, n};
                    break;
                }
                const uint32_t b = d->sym_text_off[sl.val];
                if (sl.key == h && sl.pad == n && std::memcmp(d->sym_text + b, d->sym_text + a, n * 4) == 0) {
                    mc_set_error("symbols %u and %u have the same text", sl.val, s);
                    delete t;
                    return MC_ERR_TABLE;
                }
            }
        }
    }

    // the two-choice table of the same map: cuckoo insertion; a table that will not take every symbol within a few
    // hundred moves is rebuilt at twice the size (a quarter full to begin with: in practice the first one does)
    if (d->sym_text_off && !t->wordhash.empty()) {
        auto hash_of = [&](uint32_t s) {
            const uint32_t a = d->sym_text_off[s];
            return mc_text_hash(d->sym_text + a, d->sym_text_off[s + 1] - a);
        };
        uint64_t ccap = pow2ceil(std::max<uint64_t>(16, 4ull * d->n_sym));
        for (int attempt = 0; attempt < 5; ++attempt, ccap *= 2) {
            const uint64_t mask = ccap - 1;
            t->wordck.assign(ccap, ~0ull);
            bool ok = true;
            for (uint32_t s0 = 0; s0 < d->n_sym && ok; ++s0) {
                if (d->sym_text_off[s0 + 1] == d->sym_text_off[s0]) continue;
                uint32_t s = s0;
                uint64_t h = hash_of(s);
                auto item_of = [&](uint32_t sym, uint64_t hh) {
                    return ((uint64_t)MC_CK_TAG(hh, d->sym_text_off[sym + 1] - d->sym_text_off[sym]) << 32) | sym;
                };
                const uint64_t p1 = MC_CK_POS1(h) & mask, p2 = MC_CK_POS2(h) & mask;
                if (t->wordck[p1] == ~0ull) {
                    t->wordck[p1] = item_of(s, h);
                    continue;
                }
                if (t->wordck[p2] == ~0ull) {
                    t->wordck[p2] = item_of(s, h);
                    continue;
                }
                // take a slot; whoever sat there moves to its own other slot, and so on
                uint64_t pos = p1;
                ok = false;
                for (int moves = 0; moves < 500; ++moves) {
                    const uint32_t victim = (uint32_t)t->wordck[pos];
                    t->wordck[pos] = item_of(s, h);
                    s = victim;
                    h = hash_of(s);
                    pos = pos == (MC_CK_POS1(h) & mask) ? (MC_CK_POS2(h) & mask) : (MC_CK_POS1(h) & mask);
                    if (t->wordck[pos] == ~0ull) {
                        t->wordck[pos] = item_of(s, h);
                        ok = true;
                        break;
                    }
                }
            }
            if (ok) break;
            t->wordck.clear();
        }
    }

    if (bpe) {
        const uint32_t n_struct = (uint32_t)d->row_off[1];
        t->seg_off.assign(d->seg_off, d->seg_off + n_struct + 1);
        t->seg_row.assign(d->seg_row, d->seg_row + d->seg_off[n_struct]);
        t->seg_len.assign(d->seg_len, d->seg_len + d->seg_off[n_struct]);
        t->lab_off.assign(d->lab_off, d->lab_off + (d->n_labels ? d->n_labels + 1 : 0));
        t->lab_cand_off.assign(d->lab_cand_off, d->lab_cand_off + (d->n_labels ? d->n_labels + 1 : 0));
        if (d->n_labels) {
            t->lab_units.assign(d->lab_units, d->lab_units + d->lab_off[d->n_labels]);
            t->lab_cand.assign(d->lab_cand, d->lab_cand + d->lab_cand_off[d->n_labels]);
        }
        t->np_len.assign(d->np_len, d->np_len + d->n_np);
        t->np_off.assign(d->np_off, d->np_off + (d->n_np ? d->n_np + 1 : 0));
        if (d->n_np) t->np_cand.assign(d->np_cand, d->np_cand + d->np_off[d->n_np]);
        uint64_t lcap = pow2ceil(std::max<uint64_t>(16, 2ull * d->n_labels));
        t->labhash.assign(lcap, mc_slot{~0ull, 0, 0});
        for (uint32_t j = 0; j < d->n_labels; ++j) {
            const uint32_t a = d->lab_off[j], n = d->lab_off[j + 1] - a;
            const uint64_t h = mc_ctx_hash(d->lab_units + a, n);
            for (uint64_t i = h & (lcap - 1);; i = (i + 1) & (lcap - 1)) {
                mc_slot &sl = t->labhash[i];
                if (sl.key == ~0ull) {
                    sl = mc_slot{h, j, n};
                    break;
                }
                // equal hashes are told apart by the stored units on the device; equal labels are an error
                const uint32_t b = d->lab_off[sl.val];
                if (sl.key == h && sl.pad == n && std::memcmp(d->lab_units + b, d->lab_units + a, n * 4) == 0) {
                    mc_set_error("labels %u and %u are the same", sl.val, j);
                    delete t;
                    return MC_ERR_TABLE;
                }
            }
        }
    }

    // ---- (row, symbol) -> entry hash ------------------------------------------------------
    uint64_t cap = pow2ceil(std::max<uint64_t>(16, 2ull * d->n_ent));
    t->trans.assign(cap, mc_slot{~0ull, 0, 0});
    for (uint32_t r = 0; r < d->n_ctx; ++r)
        for (uint64_t e = d->row_off[r]; e < d->row_off[r + 1]; ++e) {
            uint64_t key = ((uint64_t)r << 32) | d->ent_sym[e];
            for (uint64_t i = mc_trans_hash(key) & (cap - 1);; i = (i + 1) & (cap - 1)) {
                mc_slot &s = t->trans[i];
                if (s.key == ~0ull) {
                    s = mc_slot{key, (uint32_t)e, 0};
                    break;
                }
                if (s.key == key) {
                    mc_set_error("row %u lists symbol %u twice", r, d->ent_sym[e]);
                    delete t;
                    return MC_ERR_TABLE;
                }
            }
        }
    // ---- dense resolved transitions (scoring of small single-character alphabets) ------------
    if ((d->kind == MC_KIND_NGRAM || d->kind == MC_KIND_BACKOFF) && !t->multichar && !(d->flags & MC_FLAG_NO_DENSE) &&
        (uint64_t)d->n_ctx * d->n_sym <= (1ull << 22)) {
        const uint32_t S = d->n_sym;
        t->dense.assign((size_t)d->n_ctx * S, mc_payload{0.0, MC_NONE, MC_NONE});
        // a row's own entries first, then whatever its parent resolves (parents have shorter contexts)
        std::vector<uint32_t> order(d->n_ctx);
        for (uint32_t r = 0; r < d->n_ctx; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return t->state[a].ctx_len < t->state[b].ctx_len; });
        for (uint32_t r : order) {
            mc_payload *row = t->dense.data() + (size_t)r * S;
            const uint32_t par = t->state[r].parent;
            if (par != MC_NONE) std::copy(t->dense.begin() + (size_t)par * S, t->dense.begin() + (size_t)(par + 1) * S, row);
            for (uint64_t e = d->row_off[r]; e < d->row_off[r + 1]; ++e) {
                const mc_payload &p = t->payload[e];
                row[d->ent_sym[e]] = mc_payload{p.ml2p, p.next, (uint32_t)e};
            }
        }
    }
    *out = t;
    return MC_OK;
}

extern "C" void mc_table_free(mc_table *t) { delete t; }

extern "C" int64_t mc_table_array(const mc_table *t, int what, void *dst, size_t cap) {
    if (!t) return MC_ERR_ARG;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case MC_ARR_STATE: src = t->state.data(); bytes = t->state.size() * sizeof(mc_state_rec); break;
        case MC_ARR_DRAW: src = t->draw.data(); bytes = t->draw.size() * sizeof(mc_draw_rec); break;
        case MC_ARR_SPAY: src = t->spay.data(); bytes = t->spay.size() * sizeof(mc_draw_rec); break;
        case MC_ARR_RCUM: src = t->rcum.data(); bytes = t->rcum.size() * sizeof(double); break;
        case MC_ARR_WORDHASH: src = t->wordhash.data(); bytes = t->wordhash.size() * sizeof(mc_slot); break;
        case MC_ARR_SYMTEXTOFF: src = t->sym_text_off.data(); bytes = t->sym_text_off.size() * sizeof(uint32_t); break;
        case MC_ARR_SYMTEXT: src = t->sym_text.data(); bytes = t->sym_text.size() * sizeof(uint32_t); break;
        case MC_ARR_DENSE: src = t->dense.data(); bytes = t->dense.size() * sizeof(mc_payload); break;
        case MC_ARR_WORDCK: src = t->wordck.data(); bytes = t->wordck.size() * sizeof(uint64_t); break;
        case MC_ARR_CUM: src = t->cum.data(); bytes = t->cum.size() * sizeof(double); break;
        case MC_ARR_PAYLOAD: src = t->payload.data(); bytes = t->payload.size() * sizeof(mc_payload); break;
        case MC_ARR_TRANS: src = t->trans.data(); bytes = t->trans.size() * sizeof(mc_slot); break;
        case MC_ARR_CTXHASH: src = t->ctxhash.data(); bytes = t->ctxhash.size() * sizeof(mc_slot); break;
        case MC_ARR_SYMCHARS: src = t->sym_chars.data(); bytes = t->sym_chars.size() * sizeof(uint32_t); break;
        default: mc_set_error("mc_table_array: unknown array %d", what); return MC_ERR_ARG;
    }
    if (dst && cap) std::memcpy(dst, src, std::min(cap, bytes));
    return (int64_t)bytes;
}

extern "C" int mc_table_info(const mc_table *t, uint64_t info[16]) {
    if (!t || !info) return MC_ERR_ARG;
    std::memset(info, 0, 16 * sizeof(uint64_t));
    info[MC_INFO_NCTX] = t->n_ctx;
    info[MC_INFO_NENT] = t->n_ent;
    info[MC_INFO_NSYM] = t->n_sym;
    info[MC_INFO_START_STATE] = t->start_state;
    info[MC_INFO_ROOT] = t->root;
    info[MC_INFO_TRANS_CAP] = t->trans.size();
    info[MC_INFO_CTX_CAP] = t->ctxhash.size();
    info[MC_INFO_GUIDE_LEN] = t->draw.size();
    info[MC_INFO_DEVICE_BYTES] = mc_table_device_bytes(t);
    info[MC_INFO_MAX_ROW] = t->max_row;
    info[MC_INFO_KIND] = t->kind;
    info[MC_INFO_MULTICHAR] = (t->multichar ? 1 : 0) | (t->len_in_rec ? 2 : 0) | ((uint64_t)t->max_word_len << 8);
    info[MC_INFO_START_DRAW] = t->start_draw | ((uint64_t)t->start_gbits << 32);
    info[MC_INFO_DIRECT_BUCKETS] = t->n_direct;
    info[MC_INFO_ONE_BUCKETS] = t->n_one;
    info[MC_INFO_MULTI_BUCKETS] = t->n_multi;
    return MC_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t mc_table_device_bytes(const mc_table *t) {
    return align256(t->state.size() * sizeof(mc_state_rec)) + align256(t->draw.size() * sizeof(mc_draw_rec)) +
           align256(t->spay.size() * sizeof(mc_draw_rec)) + align256(t->rcum.size() * sizeof(double)) +
           align256(t->payload.size() * sizeof(mc_payload)) +
           align256(t->trans.size() * sizeof(mc_slot)) + align256(t->ctxhash.size() * sizeof(mc_slot)) +
           align256(t->sym_chars.size() * sizeof(uint32_t)) + align256(t->wordhash.size() * sizeof(mc_slot)) +
           align256(t->wordck.size() * sizeof(uint64_t)) +
           align256(t->sym_text_off.size() * sizeof(uint32_t)) + align256(t->sym_text.size() * sizeof(uint32_t)) +
           align256((size_t)MC_STATUS_SLOTS * MC_STATUS_STRIDE * sizeof(uint64_t)) + align256(t->dense.size() * sizeof(mc_payload)) +
           align256(t->seg_off.size() * 4) + align256(t->seg_row.size() * 4) + align256(t->seg_len.size() * 4) +
           align256(t->labhash.size() * sizeof(mc_slot)) + align256(t->lab_off.size() * 4) +
           align256(t->lab_units.size() * 4) + align256(t->lab_cand_off.size() * 4) + align256(t->lab_cand.size() * 4) +
           align256(t->np_len.size() * 4) + align256(t->np_off.size() * 4) + align256(t->np_cand.size() * 4);
}
