// Test-file ingest on the host, in C++: the counting loop of the reference's
//   lib4mc/MonteCarloParent.py:56-60   (strip "\r\n", count duplicates in a dict)
// The dict is insertion-ordered, so what matters is ORDER: the unique lines come out in first-seen
// order with their multiplicities, as one code-point array + offsets -- the form mc_score (after the
// caller's code point -> symbol id map) and mc_score_words take.
// Input is the decoded text of the test file as UTF-8; lines end at '\n' (Python's text mode has
// already translated the newlines), and '\r' / '\n' are stripped from both ends of a line as
// str.strip("\r\n") does.  A last line without '\n' counts; text ending in '\n' adds no empty line.
#include "mc_internal.h"

#include <cstdio>
#include <charconv>
#include <cstring>

namespace {

inline uint64_t load_tail(const unsigned char *p, size_t n) {
    uint64_t v = 0;
    std::memcpy(&v, p, n);
    return v;
}

inline uint64_t mix64(uint64_t a, uint64_t b) {
    const __uint128_t m = (__uint128_t)(a ^ 0x9E3779B97F4A7C15ull) * (b ^ 0xE7037ED1A0B428DBull);
    return (uint64_t)m ^ (uint64_t)(m >> 64);
}

uint64_t hash_bytes(const unsigned char *p, size_t n) {
    uint64_t h = 0xA0761D6478BD642Full ^ n;
    while (n >= 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        h = mix64(h, v);
        p += 8;
        n -= 8;
    }
    h = mix64(h, load_tail(p, n));
    return h;
}

struct Slot {
    uint64_t hash;
    uint64_t at;    // the line's bytes in the arena of unique lines (compact, so it stays in cache)
    uint32_t idx;   // unique line + 1, 0 = empty
    uint32_t len;   // bytes
};

}  // namespace

struct mc_lines {
    std::vector<uint64_t> count;   // per unique line
    std::vector<uint64_t> off;     // n + 1 offsets into cps
    std::vector<uint32_t> cps;
    uint64_t info[4];              // unique lines, lines, longest unique line (code points), code points
};

extern "C" int mc_lines_ingest(const char *utf8, size_t bytes, mc_lines **out) {
    if ((!utf8 && bytes) || !out) {
        mc_set_error("mc_lines_ingest: bad argument");
        return MC_ERR_ARG;
    }
    const unsigned char *text = (const unsigned char *)utf8;
    mc_lines *L = new mc_lines();
    std::vector<unsigned char> arena;   // bytes of the unique lines, concatenated
    std::vector<uint64_t> start;        // offset of each unique line in the arena
    std::vector<uint32_t> blen;
    std::vector<Slot> tab(1 << 12);
    size_t mask = tab.size() - 1;
    uint64_t n_lines = 0;
    size_t at = 0;
    while (at < bytes) {
        const void *hit = std::memchr(text + at, '\n', bytes - at);
        const size_t nl = hit ? (size_t)((const unsigned char *)hit - text) : bytes;
        size_t a = at, b = nl;
        at = nl + 1;
        while (a < b && text[a] == '\r') ++a;
        while (b > a && text[b - 1] == '\r') --b;
        ++n_lines;
        if (b - a > 0xFFFFFFFFull) {
            delete L;
            mc_set_error("mc_lines_ingest: a line of more than 2^32 bytes");
            return MC_ERR_ARG;
        }
        const uint32_t n = (uint32_t)(b - a);
        const uint64_t h = hash_bytes(text + a, n);
        size_t i = h & mask;
        for (;; i = (i + 1) & mask) {
            Slot &s = tab[i];
            if (!s.idx) {
                s.hash = h;
                s.len = n;
                s.idx = (uint32_t)start.size() + 1;
                s.at = arena.size();
                start.push_back(arena.size());
                arena.insert(arena.end(), text + a, text + b);
                blen.push_back(n);
                L->count.push_back(1);
                if (start.size() * 2 > tab.size()) {
                    std::vector<Slot> big(tab.size() * 4);
                    const size_t m2 = big.size() - 1;
                    for (const Slot &o : tab)
                        if (o.idx) {
                            size_t j = o.hash & m2;
                            while (big[j].idx) j = (j + 1) & m2;
                            big[j] = o;
                        }
                    tab.swap(big);
                    mask = m2;
                }
                break;
            }
            if (s.hash == h && s.len == n && std::memcmp(arena.data() + s.at, text + a, n) == 0) {
                ++L->count[s.idx - 1];
                break;
            }
        }
    }
    // the unique lines as code points
    const size_t n_uniq = start.size();
    L->off.resize(n_uniq + 1);
    size_t total_bytes = 0;
    for (uint32_t n : blen) total_bytes += n;
    L->cps.reserve(total_bytes);
    uint64_t longest = 0;
    for (size_t u = 0; u < n_uniq; ++u) {
        L->off[u] = L->cps.size();
        const unsigned char *p = arena.data() + start[u];
        const size_t n = blen[u];
        for (size_t i = 0; i < n;) {
            const unsigned char c = p[i];
            uint32_t cp;
            int len;
            if (c < 0x80) { cp = c; len = 1; }
            else if ((c >> 5) == 6) { cp = c & 31; len = 2; }
            else if ((c >> 4) == 14) { cp = c & 15; len = 3; }
            else if ((c >> 3) == 30) { cp = c & 7; len = 4; }
            else len = 0;
            bool ok = len && i + len <= n;
            for (int k = 1; ok && k < len; ++k) {
                const unsigned char d = p[i + k];
                ok = (d >> 6) == 2;
                cp = (cp << 6) | (d & 63);
            }
            if (!ok) {
                delete L;
                mc_set_error("mc_lines_ingest: the text is not valid UTF-8");
                return MC_ERR_ARG;
            }
            L->cps.push_back(cp);
            i += len;
        }
        const uint64_t chars = L->cps.size() - L->off[u];
        if (chars > longest) longest = chars;
    }
    L->off[n_uniq] = L->cps.size();
    L->info[0] = n_uniq;
    L->info[1] = n_lines;
    L->info[2] = longest;
    L->info[3] = L->cps.size();
    *out = L;
    return MC_OK;
}

extern "C" void mc_lines_free(mc_lines *l) { delete l; }

extern "C" int64_t mc_lines_array(const mc_lines *l, int what, void *dst, size_t cap) {
    if (!l) return MC_ERR_ARG;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case MC_LINES_COUNT: src = l->count.data(); bytes = l->count.size() * 8; break;
        case MC_LINES_OFF: src = l->off.data(); bytes = l->off.size() * 8; break;
        case MC_LINES_CPS: src = l->cps.data(); bytes = l->cps.size() * 4; break;
        case MC_LINES_INFO: src = l->info; bytes = sizeof l->info; break;
        default: mc_set_error("mc_lines_array: unknown array %d", what); return MC_ERR_ARG;
    }
    if (dst && cap) std::memcpy(dst, src, std::min(cap, bytes));
    return (int64_t)bytes;
}

// ---- result writer ---------------------------------------------------------------------------------
// The formatting loop of MonteCarloLib.write2 (lib4mc/MonteCarloLib.py:65-68):
//     f"{pwd}\t{mlp:.8f}\t{appearance}\t{rank}\t{cracked}\t{cracked_ratio:5.2f}\n"
// for n passwords given as one '\n'-separated UTF-8 text.  printf's %.8f / %5.2f round the exact binary
// value to nearest-even like Python's format().  Returns the bytes written, or MC_ERR_WORKSPACE when `cap`
// is too small (len(text) + 120 * n always suffices).
extern "C" int64_t mc_format_gc(const char *pwds_utf8, size_t bytes, uint64_t n, const double *mlp,
                                const int64_t *appearance, const int64_t *rank, const int64_t *cracked,
                                const double *ratio, char *dst, size_t cap) {
    if ((bytes && !pwds_utf8) || (n && (!mlp || !appearance || !rank || !cracked || !ratio || !dst))) {
        mc_set_error("mc_format_gc: bad argument");
        return MC_ERR_ARG;
    }
    size_t at = 0, out = 0;
    for (uint64_t i = 0; i < n; ++i) {
        size_t end = at;
        while (end < bytes && pwds_utf8[end] != '\n') ++end;
        if (i + 1 < n && end >= bytes) {
            mc_set_error("mc_format_gc: %llu passwords announced, the text holds %llu", (unsigned long long)n,
                         (unsigned long long)(i + 1));
            return MC_ERR_ARG;
        }
        const size_t len = end - at;
        if (out + len + 120 > cap) {
            mc_set_error("mc_format_gc: output buffer of %zu bytes is too small", cap);
            return MC_ERR_WORKSPACE;
        }
        std::memcpy(dst + out, pwds_utf8 + at, len);
        out += len;
        // std::to_chars: correctly rounded like printf / Python's format(), and independent of the process
        // locale (a host that called setlocale(LC_NUMERIC, "de_DE") would get decimal commas from printf)
        char *p = dst + out, *const lim = dst + out + 119;
        bool ok = true;
        auto put_f = [&](double v, int prec, int width) {
            *p++ = '\t';
            char tmp[400];
            auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed, prec);
            const size_t m = (size_t)(r.ptr - tmp);
            const size_t pad = m < (size_t)width ? (size_t)width - m : 0;
            if (r.ec != std::errc() || p + pad + m > lim) { ok = false; return; }
            std::memset(p, ' ', pad);
            std::memcpy(p + pad, tmp, m);
            p += pad + m;
        };
        auto put_i = [&](long long v) {
            *p++ = '\t';
            auto r = std::to_chars(p, lim, v);
            if (r.ec != std::errc()) { ok = false; return; }
            p = r.ptr;
        };
        put_f(mlp[i], 8, 0);
        if (ok) put_i((long long)appearance[i]);
        if (ok) put_i((long long)rank[i]);
        if (ok) put_i((long long)cracked[i]);
        if (ok) put_f(ratio[i], 2, 5);
        if (!ok) {
            mc_set_error("mc_format_gc: a line's numbers do not fit 120 characters");
            return MC_ERR_ARG;
        }
        *p++ = '\n';
        out = (size_t)(p - dst);
        at = end + 1;
    }
    return (int64_t)out;
}
