// Trainers on the host, in C++: the counting and smoothing of the reference's
//   nwords/nwords_trainer.py:22-53            (fixed-order n-gram)
//   backwords/backwords_trainer.py:23-78      (backoff)
// producing the flattened rows mc_table_compile takes, without a Python dict in between.
//
// What defines the result is ORDER and ARITHMETIC, both restated exactly:
//   * every dict of the reference is insertion-ordered: unique lines in first-seen order (grouped by
//     length in first-seen length order for the backoff trainer), contexts in first-seen order of the
//     scan, transitions in first-seen order, parent-only symbols appended in the parent's order;
//   * p = count / total is one correctly rounded division; `sum(...)` adds left to right starting
//     from the integer 0; `kept.get(s, 0) + p * missing` is a rounded product then a rounded sum
//     (this file is compiled with -ffp-contract=off); -log2 p is libm's log2, as math.log2 is;
//     the cumulative sums are numpy's sequential fp64 cumsum.
// Input is the decoded text of the training file as UTF-8 with '\n' line ends (Python's text mode
// has already translated the newlines); symbols are Unicode code points when splitter == 0, else
// the columns between occurrences of the one-character splitter, taken first::step.
#include "mc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>

namespace {

typedef std::vector<uint32_t> Seq;

struct SeqHash {
    size_t operator()(const Seq &s) const {
        uint64_t h = 0xCBF29CE484222325ull ^ s.size();
        for (uint32_t v : s) h = (h ^ v) * 0x100000001B3ull;
        return (size_t)(h ^ (h >> 29));
    }
};

// symbol text (as code points) <-> id, ids in first-seen order
struct Interner {
    std::unordered_map<Seq, uint32_t, SeqHash> id;
    std::vector<Seq> text;
    uint32_t get(const Seq &t) {
        auto it = id.find(t);
        if (it != id.end()) return it->second;
        const uint32_t i = (uint32_t)text.size();
        id.emplace(t, i);
        text.push_back(t);
        return i;
    }
};

// an insertion-ordered {symbol: count}
struct Row {
    Seq ctx;
    std::vector<uint32_t> sym;
    std::vector<uint64_t> cnt;
};

// insertion-ordered {context: Row}; open-addressed indexes (no allocation per lookup): contexts by
// FNV of their symbols, (row, symbol) -> position in the row by a mixed 64-bit key
struct RowTable {
    std::vector<Row> rows;
    std::vector<uint32_t> ctx_slot;              // row + 1, 0 = empty
    std::vector<uint64_t> pos_key;               // row << 32 | sym, ~0 = empty
    std::vector<uint32_t> pos_val;
    size_t n_pos = 0;

    static uint64_t hash_ctx(const uint32_t *c, uint32_t n) {
        uint64_t h = 0xCBF29CE484222325ull ^ n;
        for (uint32_t i = 0; i < n; ++i) h = (h ^ c[i]) * 0x100000001B3ull;
        return h ^ (h >> 31);
    }
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void grow_ctx() {
        std::vector<uint32_t> fresh(ctx_slot.empty() ? 1024 : ctx_slot.size() * 2, 0);
        const size_t mask = fresh.size() - 1;
        for (uint32_t r = 0; r < rows.size(); ++r) {
            size_t i = hash_ctx(rows[r].ctx.data(), (uint32_t)rows[r].ctx.size()) & mask;
            while (fresh[i]) i = (i + 1) & mask;
            fresh[i] = r + 1;
        }
        ctx_slot.swap(fresh);
    }
    void grow_pos() {
        std::vector<uint64_t> k(pos_key.empty() ? 4096 : pos_key.size() * 2, ~0ull);
        std::vector<uint32_t> v(k.size(), 0);
        const size_t mask = k.size() - 1;
        for (size_t j = 0; j < pos_key.size(); ++j)
            if (pos_key[j] != ~0ull) {
                size_t i = mix(pos_key[j]) & mask;
                while (k[i] != ~0ull) i = (i + 1) & mask;
                k[i] = pos_key[j];
                v[i] = pos_val[j];
            }
        pos_key.swap(k);
        pos_val.swap(v);
    }
    // MC_NONE when the context has no row
    uint32_t find(const uint32_t *ctx, uint32_t n) const {
        if (ctx_slot.empty()) return MC_NONE;
        const size_t mask = ctx_slot.size() - 1;
        for (size_t i = hash_ctx(ctx, n) & mask; ctx_slot[i]; i = (i + 1) & mask) {
            const Seq &have = rows[ctx_slot[i] - 1].ctx;
            if (have.size() == n && (n == 0 || std::memcmp(have.data(), ctx, n * sizeof(uint32_t)) == 0))
                return ctx_slot[i] - 1;
        }
        return MC_NONE;
    }
    uint32_t row_of(const uint32_t *ctx, uint32_t n) {
        if ((rows.size() + 1) * 2 > ctx_slot.size()) grow_ctx();
        const size_t mask = ctx_slot.size() - 1;
        size_t i = hash_ctx(ctx, n) & mask;
        for (; ctx_slot[i]; i = (i + 1) & mask) {
            const Seq &have = rows[ctx_slot[i] - 1].ctx;
            if (have.size() == n && (n == 0 || std::memcmp(have.data(), ctx, n * sizeof(uint32_t)) == 0))
                return ctx_slot[i] - 1;
        }
        const uint32_t r = (uint32_t)rows.size();
        ctx_slot[i] = r + 1;
        rows.emplace_back();
        rows.back().ctx.assign(ctx, ctx + n);
        return r;
    }
    void add(uint32_t r, uint32_t s, uint64_t c) {
        if ((n_pos + 1) * 2 > pos_key.size()) grow_pos();
        const uint64_t key = ((uint64_t)r << 32) | s;
        const size_t mask = pos_key.size() - 1;
        size_t i = mix(key) & mask;
        for (; pos_key[i] != ~0ull; i = (i + 1) & mask)
            if (pos_key[i] == key) {
                rows[r].cnt[pos_val[i]] += c;
                return;
            }
        pos_key[i] = key;
        pos_val[i] = (uint32_t)rows[r].sym.size();
        ++n_pos;
        rows[r].sym.push_back(s);
        rows[r].cnt.push_back(c);
    }
};

struct Corpus {
    Interner syms;
    std::vector<Seq> uniq;          // unique section tuples, first-seen order
    std::vector<uint64_t> mult;
    std::vector<uint32_t> word_sym;  // `words`: symbol -> occurrences, first-seen order
    std::vector<uint64_t> word_cnt;
    std::unordered_map<uint32_t, uint32_t> word_pos;
    uint32_t start_sym, end_sym;
};

bool decode_utf8(const char *p, size_t n, Seq &out) {
    out.clear();
    for (size_t i = 0; i < n;) {
        const unsigned char c = (unsigned char)p[i];
        uint32_t cp;
        int len;
        if (c < 0x80) { cp = c; len = 1; }
        else if ((c >> 5) == 6) { cp = c & 31; len = 2; }
        else if ((c >> 4) == 14) { cp = c & 15; len = 3; }
        else if ((c >> 3) == 30) { cp = c & 7; len = 4; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; ++k) {
            const unsigned char d = (unsigned char)p[i + k];
            if ((d >> 6) != 2) return false;
            cp = (cp << 6) | (d & 63);
        }
        // (overlong forms and values beyond U+10FFFF are not UTF-8 either; surrogates pass, as Python's "surrogatepass")
        if ((len == 2 && cp < 0x80u) || (len == 3 && cp < 0x800u) || (len == 4 && (cp < 0x10000u || cp > 0x10FFFFu))) return false;
        out.push_back(cp);
        i += len;
    }
    return true;
}

// tally_lines: sections = lead x start + columns + end, per line (nwords_trainer.py:30-38,
// backwords_trainer.py:31-40)
int read_corpus(const char *utf8, size_t bytes, uint32_t splitter, uint32_t first, uint32_t step, uint32_t n_lead,
                uint32_t start_cp, uint32_t end_cp, Corpus &c) {
    if (step == 0) {
        mc_set_error("trainer: skip4word must be at least 1");
        return MC_ERR_ARG;
    }
    c.start_sym = c.syms.get(Seq{start_cp});
    c.end_sym = c.syms.get(Seq{end_cp});
    // (symbols and `words` persist over calls on one Corpus; the unique lines are per call)
    c.uniq.clear();
    c.mult.clear();
    std::unordered_map<Seq, uint32_t, SeqHash> seen;
    std::unordered_map<uint32_t, uint32_t> &word_pos = c.word_pos;
    Seq line, secs, col;
    size_t at = 0;
    while (at < bytes) {
        size_t nl = at;
        while (nl < bytes && utf8[nl] != '\n') ++nl;
        size_t a = at, b = nl;
        at = nl + 1;
        while (a < b && (utf8[a] == '\r' || utf8[a] == '\n')) ++a;   // line.strip("\r\n")
        while (b > a && (utf8[b - 1] == '\r' || utf8[b - 1] == '\n')) --b;
        if (!decode_utf8(utf8 + a, b - a, line)) {
            mc_set_error("trainer: the training text is not valid UTF-8");
            return MC_ERR_ARG;
        }
        secs.assign(n_lead, c.start_sym);
        if (splitter == 0) {
            for (uint32_t cp : line) secs.push_back(c.syms.get(Seq{cp}));
        } else {
            // re.split(one literal character): the pieces between occurrences, empty ones included
            uint32_t k = 0;
            col.clear();
            for (size_t i = 0; i <= line.size(); ++i) {
                if (i == line.size() || line[i] == splitter) {
                    if (k >= first && (k - first) % step == 0) secs.push_back(c.syms.get(col));
                    ++k;
                    col.clear();
                } else {
                    col.push_back(line[i]);
                }
            }
        }
        secs.push_back(c.end_sym);
        for (uint32_t s : secs) {
            auto it = word_pos.find(s);
            if (it == word_pos.end()) {
                word_pos.emplace(s, (uint32_t)c.word_sym.size());
                c.word_sym.push_back(s);
                c.word_cnt.push_back(1);
            } else {
                ++c.word_cnt[it->second];
            }
        }
        auto it = seen.find(secs);
        if (it == seen.end()) {
            seen.emplace(secs, (uint32_t)c.uniq.size());
            c.uniq.push_back(secs);
            c.mult.push_back(1);
        } else {
            ++c.mult[it->second];
        }
    }
    return MC_OK;
}

// transitions(): (context of `order` symbols -> next symbol) over every position of the given lines
void count_transitions(const Corpus &c, const std::vector<uint32_t> &lines, uint32_t order, RowTable &t) {
    for (uint32_t li : lines) {
        const Seq &s = c.uniq[li];
        for (size_t i = order; i < s.size(); ++i) {
            const uint32_t r = t.row_of(s.data() + (i - order), order);
            t.add(r, s[i], c.mult[li]);
        }
    }
}

}  // namespace

struct mc_trained {
    std::vector<uint32_t> ctx_off, ctx_units, ent_sym, sym_text_off, sym_text, word_sym;
    std::vector<uint64_t> row_off, word_cnt;
    std::vector<double> ent_prob, ent_cum, ent_ml2p;
    uint32_t start_sym, end_sym, n_lead;
    bool v2 = false;  // contexts are code points, not symbol ids (mc_train_v2)
};

namespace {

struct ProbRow {
    Seq ctx;
    std::vector<uint32_t> sym;
    std::vector<double> p;
};

mc_trained *finish(const Corpus &c, const std::vector<ProbRow> &rows, uint32_t n_lead) {
    mc_trained *t = new mc_trained();
    t->start_sym = c.start_sym;
    t->end_sym = c.end_sym;
    t->n_lead = n_lead;
    t->ctx_off.push_back(0);
    t->row_off.push_back(0);
    for (const ProbRow &r : rows) {
        if (r.sym.empty()) continue;  // expand_2d drops empty rows (ProbLib.py:14-15)
        t->ctx_units.insert(t->ctx_units.end(), r.ctx.begin(), r.ctx.end());
        t->ctx_off.push_back((uint32_t)t->ctx_units.size());
        double run = 0.0;
        for (size_t j = 0; j < r.sym.size(); ++j) {
            t->ent_sym.push_back(r.sym[j]);
            t->ent_prob.push_back(r.p[j]);
            run = j == 0 ? r.p[j] : run + r.p[j];  // numpy.cumsum: sequential fp64
            t->ent_cum.push_back(run);
            t->ent_ml2p.push_back(-std::log2(r.p[j]));
        }
        t->row_off.push_back(t->ent_sym.size());
    }
    t->sym_text_off.push_back(0);
    for (const Seq &s : c.syms.text) {
        t->sym_text.insert(t->sym_text.end(), s.begin(), s.end());
        t->sym_text_off.push_back((uint32_t)t->sym_text.size());
    }
    t->word_sym = c.word_sym;
    t->word_cnt = c.word_cnt;
    return t;
}

}  // namespace

// nwords_counter (nwords/nwords_trainer.py:22-53)
extern "C" int mc_train_nwords(const char *utf8, size_t bytes, uint32_t n, uint32_t splitter_cp, uint32_t start4word,
                               uint32_t skip4word, uint32_t start_cp, uint32_t end_cp, mc_trained **out) {
    if (!out || (bytes && !utf8) || n < 2) {
        mc_set_error("mc_train_nwords: bad argument (n must be at least 2)");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    Corpus c;
    int rc = read_corpus(utf8, bytes, splitter_cp, start4word, skip4word, n - 1, start_cp, end_cp, c);
    if (rc != MC_OK) return rc;
    std::vector<uint32_t> all(c.uniq.size());
    for (uint32_t i = 0; i < all.size(); ++i) all[i] = i;
    RowTable counts;
    count_transitions(c, all, n - 1, counts);
    std::vector<ProbRow> rows;
    rows.reserve(counts.rows.size());
    for (const Row &r : counts.rows) {
        uint64_t total = 0;
        for (uint64_t v : r.cnt) total += v;
        ProbRow pr;
        pr.ctx = r.ctx;
        pr.sym = r.sym;
        for (uint64_t v : r.cnt) pr.p.push_back((double)v / (double)total);
        rows.push_back(std::move(pr));
    }
    *out = finish(c, rows, n - 1);
    return MC_OK;
}

// nwords_counter of the v2 model (nwords_v2/nwords_trainer_v2.py:12-56): a line is
// "pwd\tseg\ttag\tseg\ttag..."; the segments' LENGTHS cut pwd + end into pieces (:21-31), pieces seen fewer
// than `threshold` times are spelled out character by character (:38-43), the context of a transition is
// the last n-1 CHARACTERS written so far (:44-47).  Contexts are kept as code points, transitions as
// interned texts; dict orders are the reference's (first-seen).
extern "C" int mc_train_v2(const char *utf8, size_t bytes, uint32_t n, uint32_t end_cp, uint32_t threshold,
                           mc_trained **out) {
    if (!out || (bytes && !utf8) || n < 2) {
        mc_set_error("mc_train_v2: bad argument (n must be at least 2)");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    Corpus c;
    c.end_sym = c.syms.get(Seq{end_cp});
    c.start_sym = MC_NONE;
    std::unordered_map<Seq, uint32_t, SeqHash> seen;  // tuple of piece ids -> index in c.uniq
    Seq line, pieces, piece;
    std::vector<uint32_t> seg_len;
    size_t at = 0;
    while (at < bytes) {
        size_t nl = at;
        while (nl < bytes && utf8[nl] != '\n') ++nl;
        size_t a = at, b = nl;
        at = nl + 1;
        while (a < b && (utf8[a] == '\r' || utf8[a] == '\n')) ++a;
        while (b > a && (utf8[b - 1] == '\r' || utf8[b - 1] == '\n')) --b;
        if (!decode_utf8(utf8 + a, b - a, line)) {
            mc_set_error("trainer: the training text is not valid UTF-8");
            return MC_ERR_ARG;
        }
        // fields between tabs: field 0 is the password, the odd fields are its segments
        seg_len.clear();
        size_t pwd_len = line.size(), field = 0, start = 0;
        for (size_t i = 0; i <= line.size(); ++i)
            if (i == line.size() || line[i] == '\t') {
                if (field == 0) pwd_len = i;
                else if (field % 2 == 1) seg_len.push_back((uint32_t)(i - start));
                ++field;
                start = i + 1;
            }
        seg_len.push_back(1);  // raw_sections.append(end_chr)
        line.resize(pwd_len);
        line.push_back(end_cp);
        pieces.clear();
        size_t cut = 0, covered = 0;
        for (uint32_t len : seg_len) {
            const size_t lo = std::min(cut, line.size()), hi = std::min(cut + len, line.size());  // (slices clamp)
            piece.assign(line.begin() + lo, line.begin() + hi);
            cut += len;
            covered += piece.size();
            const uint32_t id = c.syms.get(piece);
            pieces.push_back(id);
            auto it = c.word_pos.find(id);
            if (it == c.word_pos.end()) {
                c.word_pos.emplace(id, (uint32_t)c.word_sym.size());
                c.word_sym.push_back(id);
                c.word_cnt.push_back(1);
            } else {
                ++c.word_cnt[it->second];
            }
        }
        // "".join(sections) != pwd: the slices are consecutive from 0 and clamp at the end of pwd, so they
        // rebuild it exactly when they reach its end (segments that run past it leave empty pieces, which
        // the reference accepts)
        if (covered != line.size() || line.size() < 4) {
            mc_set_error("error1 (nwords_trainer_v2.py:32-33: the segments do not rebuild the password, or it is "
                         "shorter than 3 characters)");
            return MC_ERR_ARG;
        }
        auto it = seen.find(pieces);
        if (it == seen.end()) {
            seen.emplace(pieces, (uint32_t)c.uniq.size());
            c.uniq.push_back(pieces);
            c.mult.push_back(1);
        } else {
            ++c.mult[it->second];
        }
    }
    const uint32_t reach = n - 1;
    RowTable counts;  // rows keyed by the context's code points
    Seq ctx, toks;
    for (size_t u = 0; u < c.uniq.size(); ++u) {
        toks.clear();
        for (uint32_t id : c.uniq[u]) {
            if (c.word_cnt[c.word_pos[id]] >= threshold) {
                toks.push_back(id);
            } else {
                const Seq text = c.syms.text[id];  // (a copy: get() below may grow the table)
                for (uint32_t cp : text) toks.push_back(c.syms.get(Seq{cp}));
            }
        }
        ctx.clear();
        for (uint32_t tok : toks) {
            const uint32_t r = counts.row_of(ctx.data(), (uint32_t)ctx.size());
            counts.add(r, tok, c.mult[u]);
            const Seq &text = c.syms.text[tok];
            ctx.insert(ctx.end(), text.begin(), text.end());
            if (ctx.size() > reach) ctx.erase(ctx.begin(), ctx.end() - reach);
        }
    }
    std::vector<ProbRow> rows;
    rows.reserve(counts.rows.size());
    for (const Row &r : counts.rows) {
        uint64_t total = 0;
        for (uint64_t v : r.cnt) total += v;
        ProbRow pr;
        pr.ctx = r.ctx;
        pr.sym = r.sym;
        for (uint64_t v : r.cnt) pr.p.push_back((double)v / (double)total);
        rows.push_back(std::move(pr));
    }
    mc_trained *t = finish(c, rows, 0);
    t->v2 = true;
    *out = t;
    return MC_OK;
}

// backwords_counter (backwords/backwords_trainer.py:23-78)
extern "C" int mc_train_backwords(const char *utf8, size_t bytes, uint32_t splitter_cp, uint32_t start4word,
                                  uint32_t skip4word, uint32_t threshold, uint32_t max_gram, uint32_t start_cp,
                                  uint32_t end_cp, mc_trained **out) {
    if (!out || (bytes && !utf8)) {
        mc_set_error("mc_train_backwords: bad argument");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    Corpus c;
    int rc = read_corpus(utf8, bytes, splitter_cp, start4word, skip4word, 1, start_cp, end_cp, c);
    if (rc != MC_OK) return rc;
    std::vector<ProbRow> rows;
    std::unordered_map<Seq, uint32_t, SeqHash> row_index;  // context -> position in `rows`
    // zero-gram row: every symbol but the start symbol, in first-seen order over the stream (:36-45)
    {
        ProbRow z;
        uint64_t total = 0;
        for (size_t i = 0; i < c.word_sym.size(); ++i)
            if (c.word_sym[i] != c.start_sym) total += c.word_cnt[i];
        for (size_t i = 0; i < c.word_sym.size(); ++i)
            if (c.word_sym[i] != c.start_sym) {
                z.sym.push_back(c.word_sym[i]);
                z.p.push_back((double)c.word_cnt[i] / (double)total);
            }
        row_index.emplace(Seq(), 0u);
        rows.push_back(std::move(z));
    }
    // lines grouped by length, groups in first-seen order (:40, :54)
    std::vector<uint32_t> lengths;
    std::unordered_map<uint32_t, uint32_t> group_of;
    std::vector<std::vector<uint32_t>> groups;
    std::vector<uint64_t> group_total;
    for (uint32_t i = 0; i < c.uniq.size(); ++i) {
        const uint32_t ln = (uint32_t)c.uniq[i].size();
        auto it = group_of.find(ln);
        uint32_t g;
        if (it == group_of.end()) {
            g = (uint32_t)groups.size();
            group_of.emplace(ln, g);
            lengths.push_back(ln);
            groups.emplace_back();
            group_total.push_back(0);
        } else {
            g = it->second;
        }
        groups[g].push_back(i);
        group_total[g] += c.mult[i];
    }
    uint32_t longest = 0;
    bool any = false;
    for (size_t g = 0; g < groups.size(); ++g)
        if (group_total[g] >= threshold) {
            longest = std::max(longest, lengths[g]);
            any = true;
        }
    if (!any) {
        mc_set_error("backoff trainer: no line length occurs %u times (the reference raises ValueError: max() of an "
                     "empty list, backwords_trainer.py:48)", threshold);
        return MC_ERR_ARG;
    }
    const uint32_t top = std::min(longest, std::max(2u, max_gram));
    std::vector<uint32_t> lines;
    for (uint32_t n = 2; n <= top; ++n) {
        lines.clear();
        for (size_t g = 0; g < groups.size(); ++g)
            if (lengths[g] >= n) lines.insert(lines.end(), groups[g].begin(), groups[g].end());
        RowTable counts;
        count_transitions(c, lines, n - 1, counts);
        for (const Row &r : counts.rows) {
            uint64_t total = 0;
            for (uint64_t v : r.cnt) total += v;
            if (total < threshold) continue;
            ProbRow pr;
            pr.ctx = r.ctx;
            std::unordered_map<uint32_t, uint32_t> at;
            double kept_sum = 0.0;
            bool first = true;
            for (size_t j = 0; j < r.sym.size(); ++j)
                if (r.cnt[j] >= threshold) {
                    const double p = (double)r.cnt[j] / (double)total;
                    at.emplace(r.sym[j], (uint32_t)pr.sym.size());
                    pr.sym.push_back(r.sym[j]);
                    pr.p.push_back(p);
                    kept_sum = first ? p : kept_sum + p;  // sum() starts from the integer 0
                    first = false;
                }
            const double missing = 1.0 - (first ? 0.0 : kept_sum);
            if (missing == 1.0) continue;
            if (missing > 0.0) {
                const Seq parent(r.ctx.begin() + 1, r.ctx.end());
                auto pit = row_index.find(parent);
                if (pit == row_index.end()) {
                    mc_set_error("backoff trainer: a row's parent context is missing (the reference raises KeyError, "
                                 "backwords_trainer.py:74)");
                    return MC_ERR_TABLE;
                }
                const ProbRow &par = rows[pit->second];
                for (size_t j = 0; j < par.sym.size(); ++j) {
                    const double add = par.p[j] * missing;
                    auto it = at.find(par.sym[j]);
                    if (it == at.end()) {
                        at.emplace(par.sym[j], (uint32_t)pr.sym.size());
                        pr.sym.push_back(par.sym[j]);
                        pr.p.push_back(add);  // 0 + p * missing
                    } else {
                        pr.p[it->second] = pr.p[it->second] + add;
                    }
                }
            }
            row_index.emplace(pr.ctx, (uint32_t)rows.size());
            rows.push_back(std::move(pr));
        }
    }
    *out = finish(c, rows, 1);
    return MC_OK;
}

// ---- secondary (incremental) training: backwords/backwords_secondary_trainer.py -----------------------
// The raw counts persist between rounds; a round merges its training lines into them
// (backwords_counter, :23-99) and probabilities are derived afterwards (freq2prob, :102-121).
struct mc_counts {
    Corpus c;        // symbols and `words` over every round
    RowTable table;  // the rows of order >= 1, in the insertion order of nwords_dict
    bool any_line = false;
};

extern "C" int mc_counts_new(mc_counts **out) {
    if (!out) return MC_ERR_ARG;
    *out = new mc_counts();
    return MC_OK;
}

extern "C" void mc_counts_free(mc_counts *k) { delete k; }

extern "C" int mc_counts_merge(mc_counts *k, const char *utf8, size_t bytes, uint32_t splitter_cp, uint32_t start4word,
                               uint32_t skip4word, uint32_t max_gram, uint32_t threshold, uint32_t start_cp,
                               uint32_t end_cp) {
    if (!k || (bytes && !utf8)) {
        mc_set_error("mc_counts_merge: bad argument");
        return MC_ERR_ARG;
    }
    if (bytes == 0) return MC_OK;  // "No passwords for training, early return!" (:35-37)
    Corpus &c = k->c;
    int rc = read_corpus(utf8, bytes, splitter_cp, start4word, skip4word, 1, start_cp, end_cp, c);
    if (rc != MC_OK) return rc;
    if (c.uniq.empty()) return MC_OK;
    k->any_line = true;
    // this round's lines grouped by length, groups in first-seen order (:56, :63)
    std::vector<uint32_t> lengths;
    std::unordered_map<uint32_t, uint32_t> group_of;
    std::vector<std::vector<uint32_t>> groups;
    uint32_t longest = 2;
    for (uint32_t i = 0; i < c.uniq.size(); ++i) {
        const uint32_t ln = (uint32_t)c.uniq[i].size();
        longest = std::max(longest, ln);
        auto it = group_of.find(ln);
        if (it == group_of.end()) {
            group_of.emplace(ln, (uint32_t)groups.size());
            lengths.push_back(ln);
            groups.emplace_back();
            groups.back().push_back(i);
        } else {
            groups[it->second].push_back(i);
        }
    }
    std::vector<uint32_t> lines;
    for (uint32_t n = 2; n <= std::min(max_gram, longest); ++n) {
        lines.clear();
        for (size_t g = 0; g < groups.size(); ++g)
            if (lengths[g] >= n) lines.insert(lines.end(), groups[g].begin(), groups[g].end());
        RowTable fresh;
        count_transitions(c, lines, n - 1, fresh);
        if (fresh.rows.empty()) break;
        for (const Row &r : fresh.rows) {
            uint32_t have = k->table.find(r.ctx.data(), (uint32_t)r.ctx.size());
            if (have == MC_NONE) {
                // a context unseen so far is admitted only with a transition at/above the threshold (:89-92)
                bool admit = false;
                for (uint64_t v : r.cnt) admit = admit || v >= threshold;
                if (!admit) continue;
                have = k->table.row_of(r.ctx.data(), (uint32_t)r.ctx.size());
            }
            for (size_t j = 0; j < r.sym.size(); ++j) k->table.add(have, r.sym[j], r.cnt[j]);
        }
    }
    return MC_OK;
}

// A count model that exists as plain arrays -- a `(counts, words, config)` pickle written by the
// reference (backwords_secondary_main.py:28-32) or by this package -- back into native counts, so
// that later rounds merge and train in C++.  Symbols keep the order given; the row of the empty
// context is implied by `words` (the caller checks that the pickle's agrees) and is not passed.
extern "C" int mc_counts_import(mc_counts *k, const uint32_t *sym_text_off, const uint32_t *sym_text, uint32_t n_sym,
                                const uint32_t *word_sym, const uint64_t *word_cnt, uint32_t n_words,
                                const uint32_t *ctx_off, const uint32_t *ctx_units, const uint64_t *row_off,
                                const uint32_t *ent_sym, const uint64_t *ent_cnt, uint32_t n_rows, uint32_t start_cp,
                                uint32_t end_cp) {
    if (!k || !sym_text_off || (n_words && (!word_sym || !word_cnt)) ||
        (n_rows && (!ctx_off || !row_off || !ent_sym || !ent_cnt))) {
        mc_set_error("mc_counts_import: bad argument");
        return MC_ERR_ARG;
    }
    if (!k->c.syms.text.empty() || !k->table.rows.empty()) {
        mc_set_error("mc_counts_import: the counts are not empty");
        return MC_ERR_ARG;
    }
    Corpus &c = k->c;
    for (uint32_t i = 0; i < n_sym; ++i) {
        const Seq t(sym_text + sym_text_off[i], sym_text + sym_text_off[i + 1]);
        if (c.syms.get(t) != i) {
            *k = mc_counts();
            mc_set_error("mc_counts_import: symbol %u repeats an earlier one", i);
            return MC_ERR_ARG;
        }
    }
    c.start_sym = c.syms.get(Seq{start_cp});
    c.end_sym = c.syms.get(Seq{end_cp});
    const uint32_t known = (uint32_t)c.syms.text.size();
    bool ok = true;
    for (uint32_t i = 0; i < n_words && ok; ++i) {
        ok = word_sym[i] < known && c.word_pos.emplace(word_sym[i], i).second;
        c.word_sym.push_back(word_sym[i]);
        c.word_cnt.push_back(word_cnt[i]);
    }
    for (uint32_t r = 0; r < n_rows && ok; ++r) {
        const uint32_t n = ctx_off[r + 1] - ctx_off[r];
        for (uint32_t j = 0; j < n; ++j) ok = ok && ctx_units[ctx_off[r] + j] < known;
        ok = ok && n > 0 && k->table.find(ctx_units + ctx_off[r], n) == MC_NONE;
        if (!ok) break;
        const uint32_t row = k->table.row_of(ctx_units + ctx_off[r], n);
        for (uint64_t e = row_off[r]; e < row_off[r + 1] && ok; ++e) {
            ok = ent_sym[e] < known;
            if (ok) k->table.add(row, ent_sym[e], ent_cnt[e]);
        }
        ok = ok && k->table.rows[row].sym.size() == row_off[r + 1] - row_off[r];  // (no symbol twice in a row)
    }
    if (!ok) {
        *k = mc_counts();
        mc_set_error("mc_counts_import: inconsistent arrays (unknown symbol id, repeated word, context or transition, "
                     "or an empty context among the rows)");
        return MC_ERR_ARG;
    }
    k->any_line = n_words > 0;
    return MC_OK;
}

namespace {
// the zero-gram row of nwords_dict: every symbol but the start symbol, counted per occurrence (:49-54)
Row zero_row(const Corpus &c) {
    Row z;
    for (size_t i = 0; i < c.word_sym.size(); ++i)
        if (c.word_sym[i] != c.start_sym) {
            z.sym.push_back(c.word_sym[i]);
            z.cnt.push_back(c.word_cnt[i]);
        }
    return z;
}
}  // namespace

// freq2prob (:102-121) -> rows ready for mc_trained_compile
extern "C" int mc_counts_train(const mc_counts *k, uint32_t threshold, mc_trained **out) {
    if (!k || !out) {
        mc_set_error("mc_counts_train: bad argument");
        return MC_ERR_ARG;
    }
    *out = nullptr;
    if (!k->any_line) {
        mc_set_error("mc_counts_train: no training line has been merged yet");
        return MC_ERR_ARG;
    }
    const Row zero = zero_row(k->c);
    std::vector<const Row *> order;
    order.push_back(&zero);
    for (const Row &r : k->table.rows) order.push_back(&r);
    std::stable_sort(order.begin(), order.end(),
                     [](const Row *a, const Row *b) { return a->ctx.size() < b->ctx.size(); });
    std::vector<ProbRow> rows;
    std::unordered_map<Seq, uint32_t, SeqHash> row_index;
    for (const Row *r : order) {
        uint64_t total = 0;
        for (uint64_t v : r->cnt) total += v;
        ProbRow pr;
        pr.ctx = r->ctx;
        std::unordered_map<uint32_t, uint32_t> at;
        double kept_sum = 0.0;
        bool first = true;
        for (size_t j = 0; j < r->sym.size(); ++j)
            if (r->cnt[j] >= threshold) {
                const double p = (double)r->cnt[j] / (double)total;
                at.emplace(r->sym[j], (uint32_t)pr.sym.size());
                pr.sym.push_back(r->sym[j]);
                pr.p.push_back(p);
                kept_sum = first ? p : kept_sum + p;
                first = false;
            }
        if (pr.sym.empty()) continue;  // every transition is below the threshold (:107-110)
        if (pr.sym.size() < r->sym.size() && !r->ctx.empty()) {
            const double missing = 1.0 - kept_sum;
            auto pit = row_index.find(Seq(r->ctx.begin() + 1, r->ctx.end()));
            if (pit == row_index.end()) {
                mc_set_error("freq2prob: a row's parent context has no probabilities (the reference raises KeyError, "
                             "backwords_secondary_trainer.py:117)");
                return MC_ERR_TABLE;
            }
            const ProbRow &par = rows[pit->second];
            for (size_t j = 0; j < par.sym.size(); ++j) {
                const double add = par.p[j] * missing;
                auto it = at.find(par.sym[j]);
                if (it == at.end()) {
                    at.emplace(par.sym[j], (uint32_t)pr.sym.size());
                    pr.sym.push_back(par.sym[j]);
                    pr.p.push_back(0.0 + add);
                } else {
                    pr.p[it->second] = pr.p[it->second] + add;
                }
            }
        }
        row_index.emplace(pr.ctx, (uint32_t)rows.size());
        rows.push_back(std::move(pr));
    }
    *out = finish(k->c, rows, 1);
    return MC_OK;
}

// the counts as flat arrays (for the Python view that gets pickled): zero-gram row first, then the rows
// in insertion order; `what` as for mc_trained_array, MC_TRAINED_ENT_PROB/CUM/ML2P replaced by
// MC_COUNTS_ENT_COUNT
extern "C" int64_t mc_counts_array(const mc_counts *k, int what, void *dst, size_t cap) {
    if (!k) return MC_ERR_ARG;
    std::vector<uint32_t> u32;
    std::vector<uint64_t> u64;
    const Row zero = zero_row(k->c);
    std::vector<const Row *> rows;
    if (k->any_line) rows.push_back(&zero);
    for (const Row &r : k->table.rows) rows.push_back(&r);
    bool wide = false;
    switch (what) {
        case MC_TRAINED_CTX_OFF:
            u32.push_back(0);
            for (const Row *r : rows) u32.push_back(u32.back() + (uint32_t)r->ctx.size());
            break;
        case MC_TRAINED_CTX_UNITS:
            for (const Row *r : rows) u32.insert(u32.end(), r->ctx.begin(), r->ctx.end());
            break;
        case MC_TRAINED_ROW_OFF:
            wide = true;
            u64.push_back(0);
            for (const Row *r : rows) u64.push_back(u64.back() + r->sym.size());
            break;
        case MC_TRAINED_ENT_SYM:
            for (const Row *r : rows) u32.insert(u32.end(), r->sym.begin(), r->sym.end());
            break;
        case MC_COUNTS_ENT_COUNT:
            wide = true;
            for (const Row *r : rows) u64.insert(u64.end(), r->cnt.begin(), r->cnt.end());
            break;
        case MC_TRAINED_SYM_TEXT_OFF:
            u32.push_back(0);
            for (const Seq &s : k->c.syms.text) u32.push_back(u32.back() + (uint32_t)s.size());
            break;
        case MC_TRAINED_SYM_TEXT:
            for (const Seq &s : k->c.syms.text) u32.insert(u32.end(), s.begin(), s.end());
            break;
        case MC_TRAINED_WORD_SYM: u32 = k->c.word_sym; break;
        case MC_TRAINED_WORD_COUNT: wide = true; u64 = k->c.word_cnt; break;
        default: mc_set_error("mc_counts_array: unknown array %d", what); return MC_ERR_ARG;
    }
    const void *src = wide ? (const void *)u64.data() : (const void *)u32.data();
    const size_t bytes = wide ? u64.size() * 8 : u32.size() * 4;
    if (dst && cap) std::memcpy(dst, src, std::min(cap, bytes));
    return (int64_t)bytes;
}

extern "C" void mc_trained_free(mc_trained *t) { delete t; }

extern "C" int64_t mc_trained_array(const mc_trained *t, int what, void *dst, size_t cap) {
    if (!t) return MC_ERR_ARG;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case MC_TRAINED_CTX_OFF: src = t->ctx_off.data(); bytes = t->ctx_off.size() * 4; break;
        case MC_TRAINED_CTX_UNITS: src = t->ctx_units.data(); bytes = t->ctx_units.size() * 4; break;
        case MC_TRAINED_ROW_OFF: src = t->row_off.data(); bytes = t->row_off.size() * 8; break;
        case MC_TRAINED_ENT_SYM: src = t->ent_sym.data(); bytes = t->ent_sym.size() * 4; break;
        case MC_TRAINED_ENT_PROB: src = t->ent_prob.data(); bytes = t->ent_prob.size() * 8; break;
        case MC_TRAINED_ENT_CUM: src = t->ent_cum.data(); bytes = t->ent_cum.size() * 8; break;
        case MC_TRAINED_ENT_ML2P: src = t->ent_ml2p.data(); bytes = t->ent_ml2p.size() * 8; break;
        case MC_TRAINED_SYM_TEXT_OFF: src = t->sym_text_off.data(); bytes = t->sym_text_off.size() * 4; break;
        case MC_TRAINED_SYM_TEXT: src = t->sym_text.data(); bytes = t->sym_text.size() * 4; break;
        case MC_TRAINED_WORD_SYM: src = t->word_sym.data(); bytes = t->word_sym.size() * 4; break;
        case MC_TRAINED_WORD_COUNT: src = t->word_cnt.data(); bytes = t->word_cnt.size() * 8; break;
        default: mc_set_error("mc_trained_array: unknown array %d", what); return MC_ERR_ARG;
    }
    if (dst && cap) std::memcpy(dst, src, std::min(cap, bytes));
    return (int64_t)bytes;
}

// The trained rows as the descriptor mc_table_compile takes (n-gram / backoff kinds).
extern "C" int mc_trained_compile(const mc_trained *t, uint32_t kind, uint32_t order, uint32_t min_len,
                                  uint32_t max_len, uint32_t flags, uint32_t guide_factor, mc_table **out) {
    if (!t || !out || (kind != MC_KIND_NGRAM && kind != MC_KIND_BACKOFF && kind != MC_KIND_V2)) {
        mc_set_error("mc_trained_compile: bad argument");
        return MC_ERR_ARG;
    }
    std::vector<uint32_t> chars(t->sym_text_off.size() - 1), start(t->n_lead, t->start_sym);
    for (size_t s = 0; s + 1 < t->sym_text_off.size(); ++s) chars[s] = t->sym_text_off[s + 1] - t->sym_text_off[s];
    mc_model_desc d;
    std::memset(&d, 0, sizeof d);
    d.kind = kind;
    d.order = order;
    d.n_ctx = (uint32_t)(t->ctx_off.size() - 1);
    d.n_sym = (uint32_t)chars.size();
    d.n_ent = t->ent_sym.size();
    d.ctx_off = t->ctx_off.data();
    d.ctx_units = t->ctx_units.data();
    d.row_off = t->row_off.data();
    d.ent_sym = t->ent_sym.data();
    d.ent_cum = t->ent_cum.data();
    d.ent_ml2p = t->ent_ml2p.data();
    d.sym_chars = chars.data();
    d.start_units = start.data();
    d.n_start = t->n_lead;
    d.end_sym = t->end_sym;
    d.min_len = min_len;
    d.max_len = max_len;
    d.flags = flags;
    d.guide_factor = guide_factor;
    d.sym_text_off = t->sym_text_off.data();
    d.sym_text = t->sym_text.data();
    if (t->v2) {
        // contexts are characters: a symbol's units are the code points of its text
        if (kind != MC_KIND_V2) {
            mc_set_error("mc_trained_compile: a model trained by mc_train_v2 compiles as MC_KIND_V2 only");
            return MC_ERR_ARG;
        }
        d.sym_unit_off = t->sym_text_off.data();
        d.sym_units = t->sym_text.data();
    } else if (kind == MC_KIND_V2) {
        mc_set_error("mc_trained_compile: MC_KIND_V2 needs a model trained by mc_train_v2");
        return MC_ERR_ARG;
    }
    return mc_table_compile(&d, out);
}
