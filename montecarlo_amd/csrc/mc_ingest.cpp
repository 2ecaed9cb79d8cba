// Test-file ingest on the host, in C++: the counting loop of the reference's
//   lib4mc/MonteCarloParent.py:56-60   (strip "\r\n", count duplicates in a dict)
// The dict is insertion-ordered, so what matters is ORDER: the unique lines come out in first-seen
// order with their multiplicities, as one code-point array + offsets -- the form mc_score (after the
// caller's code point -> symbol id map) and mc_score_words take.
// Input is the decoded text of the test file as UTF-8; lines end at '\n' (Python's text mode has
// already translated the newlines), and '\r' / '\n' are stripped from both ends of a line as
// str.strip("\r\n") does.  A last line without '\n' counts; text ending in '\n' adds no empty line.
#include "mc_internal.h"

#include <sched.h>

#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <thread>

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
    uint32_t tag;   // high half of the line's hash
    uint32_t idx;   // unique line + 1, 0 = empty
};

struct Line {
    uint64_t hash;
    size_t a;
    uint32_t n;
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
    std::vector<uint64_t> hashes;       // of each unique line (the table keeps half of it)
    // (8-byte slots: 10^6 unique lines probe a 16 MB table, not a 100 MB one; sized from the number of lines so
    // that a file of distinct lines is re-hashed once, while a file of repeats keeps a table that fits the cache)
    size_t newlines = 0;
    for (const unsigned char *q = text, *e = text + bytes; q < e; ++newlines) {
        const void *hit = std::memchr(q, '\n', (size_t)(e - q));
        if (!hit) break;
        q = (const unsigned char *)hit + 1;
    }
    size_t slots = 1 << 12;
    while (slots < newlines / 2) slots <<= 1;
    std::vector<Slot> tab(slots);
    start.reserve(newlines / 4);
    blen.reserve(newlines / 4);
    hashes.reserve(newlines / 4);
    L->count.reserve(newlines / 4);
    arena.reserve(bytes / 4);
    size_t mask = tab.size() - 1;
    uint64_t n_lines = 0;
    size_t at = 0;
    constexpr int BATCH = 16;            // lines hashed ahead of their insertion, their slots prefetched
    Line batch[BATCH];
    while (at < bytes) {
        int nb = 0;
        while (nb < BATCH && at < bytes) {
            const void *hit = std::memchr(text + at, '\n', bytes - at);
            const size_t nl = hit ? (size_t)((const unsigned char *)hit - text) : bytes;
            size_t a = at, b = nl;
            at = nl + 1;
            while (a < b && text[a] == '\r') ++a;
            while (b > a && text[b - 1] == '\r') --b;
            if (b - a > 0xFFFFFFFFull) {
                delete L;
                mc_set_error("mc_lines_ingest: a line of more than 2^32 bytes");
                return MC_ERR_ARG;
            }
            batch[nb].a = a;
            batch[nb].n = (uint32_t)(b - a);
            batch[nb].hash = hash_bytes(text + a, b - a);
            __builtin_prefetch(&tab[batch[nb].hash & mask]);
            ++nb;
        }
        n_lines += (uint64_t)nb;
        for (int q = 0; q < nb; ++q) {
            const uint64_t h = batch[q].hash;
            const size_t a = batch[q].a;
            const uint32_t n = batch[q].n, tag = (uint32_t)(h >> 32);
            for (size_t i = h & mask;; i = (i + 1) & mask) {
                Slot &s = tab[i];
                if (!s.idx) {
                    s.tag = tag;
                    s.idx = (uint32_t)start.size() + 1;
                    start.push_back(arena.size());
                    arena.insert(arena.end(), text + a, text + a + n);
                    blen.push_back(n);
                    hashes.push_back(h);
                    L->count.push_back(1);
                    if (start.size() * 2 > tab.size()) {
                        std::vector<Slot> big(tab.size() * 4);
                        const size_t m2 = big.size() - 1;
                        for (const Slot &o : tab)
                            if (o.idx) {
                                size_t j = hashes[o.idx - 1] & m2;
                                while (big[j].idx) j = (j + 1) & m2;
                                big[j] = o;
                            }
                        tab.swap(big);
                        mask = m2;
                    }
                    break;
                }
                // (n == 0: nothing to compare, and an empty arena's data() may be null -- memcmp must not be handed one)
                if (s.tag == tag && blen[s.idx - 1] == n &&
                    (n == 0 || std::memcmp(arena.data() + start[s.idx - 1], text + a, n) == 0)) {
                    ++L->count[s.idx - 1];
                    break;
                }
            }
        }
    }
    // the unique lines as code points
    const size_t n_uniq = start.size();
    L->off.resize(n_uniq + 1);
    size_t total_bytes = 0;
    for (uint32_t n : blen) total_bytes += n;
    L->cps.resize(total_bytes);            // (a code point takes at least one byte; trimmed below)
    uint32_t *const cp0 = L->cps.data();
    uint32_t *w = cp0;
    uint64_t longest = 0;
    for (size_t u = 0; u < n_uniq; ++u) {
        L->off[u] = (uint64_t)(w - cp0);
        const unsigned char *p = arena.data() + start[u];
        const size_t n = blen[u];
        for (size_t i = 0; i < n;) {
            const unsigned char c = p[i];
            if (c < 0x80) {                // (the common case, kept short)
                *w++ = c;
                ++i;
                continue;
            }
            uint32_t cp;
            int len;
            if ((c >> 5) == 6) { cp = c & 31; len = 2; }
            else if ((c >> 4) == 14) { cp = c & 15; len = 3; }
            else if ((c >> 3) == 30) { cp = c & 7; len = 4; }
            else len = 0;
            bool ok = len && i + len <= n;
            for (int k = 1; ok && k < len; ++k) {
                const unsigned char d = p[i + k];
                ok = (d >> 6) == 2;
                cp = (cp << 6) | (d & 63);
            }
            // (overlong forms and values beyond U+10FFFF are not UTF-8 either; surrogates pass: Python's
            // "surrogatepass" carries undecodable bytes of the file that way)
            if (ok && ((len == 2 && cp < 0x80u) || (len == 3 && cp < 0x800u) || (len == 4 && (cp < 0x10000u || cp > 0x10FFFFu))))
                ok = false;
            if (!ok) {
                delete L;
                mc_set_error("mc_lines_ingest: the text is not valid UTF-8");
                return MC_ERR_ARG;
            }
            *w++ = cp;
            i += len;
        }
        const uint64_t chars = (uint64_t)(w - cp0) - L->off[u];
        if (chars > longest) longest = chars;
    }
    L->cps.resize((size_t)(w - cp0));
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
namespace {
// "\t%.8f\t%lld\t%lld\t%lld\t%5.2f\n" without printf: std::to_chars is correctly rounded like printf / Python's
// format(), and independent of the process locale (a host that called setlocale(LC_NUMERIC, "de_DE") would get
// decimal commas from printf).  Writes at most 119 bytes at p; returns the new end, or nullptr when it does not fit.
char *format_gc_numbers(char *p, double mlp, long long appearance, long long rank, long long cracked, double ratio) {
    char *const lim = p + 119;
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
    put_f(mlp, 8, 0);
    if (ok) put_i(appearance);
    if (ok) put_i(rank);
    if (ok) put_i(cracked);
    if (ok) put_f(ratio, 2, 5);
    if (!ok) return nullptr;
    *p++ = '\n';
    return p;
}
}  // namespace

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
        char *e = format_gc_numbers(dst + out, mlp[i], (long long)appearance[i], (long long)rank[i],
                                    (long long)cracked[i], ratio[i]);
        if (!e) {
            mc_set_error("mc_format_gc: a line's numbers do not fit 120 characters");
            return MC_ERR_ARG;
        }
        out = (size_t)(e - dst);
        at = end + 1;
    }
    return (int64_t)out;
}

// The same lines with the passwords given as the code points of mc_lines_ingest (MC_LINES_CPS / MC_LINES_OFF)
// and an order: row i is unique line order[i].  Nothing becomes a Python string on the way.  A lone surrogate
// has no UTF-8 form: MC_ERR_ARG, and the caller writes that file with the reference's f-string loop.
// Rows are formatted in blocks by as many threads as the process may run on (at most 16): a block's bytes go to
// a buffer of its own, and once the sizes are known every block is copied to its place in dst.
namespace {

constexpr uint64_t FORMAT_BLOCK = 16384;  // rows per block

// rows [r0, r1) -> buf (large enough for the worst case); returns the bytes, -1 lone surrogate, -2 numbers too wide
int64_t format_rows_cps(const uint32_t *cps, const uint64_t *off, const uint32_t *order, uint64_t r0, uint64_t r1,
                        const double *mlp, const int64_t *appearance, const int64_t *rank, const int64_t *cracked,
                        const double *ratio, char *buf) {
    unsigned char *p = (unsigned char *)buf;
    for (uint64_t i = r0; i < r1; ++i) {
        const uint64_t a = off[order[i]], b = off[order[i] + 1];
        for (uint64_t j = a; j < b; ++j) {
            const uint32_t c = cps[j];
            if (c < 0x80) *p++ = (unsigned char)c;
            else if (c < 0x800) {
                *p++ = (unsigned char)(0xC0 | (c >> 6));
                *p++ = (unsigned char)(0x80 | (c & 0x3F));
            } else if (c < 0x10000) {
                if (c >= 0xD800 && c < 0xE000) return -1;
                *p++ = (unsigned char)(0xE0 | (c >> 12));
                *p++ = (unsigned char)(0x80 | ((c >> 6) & 0x3F));
                *p++ = (unsigned char)(0x80 | (c & 0x3F));
            } else {
                *p++ = (unsigned char)(0xF0 | (c >> 18));
                *p++ = (unsigned char)(0x80 | ((c >> 12) & 0x3F));
                *p++ = (unsigned char)(0x80 | ((c >> 6) & 0x3F));
                *p++ = (unsigned char)(0x80 | (c & 0x3F));
            }
        }
        char *e = format_gc_numbers((char *)p, mlp[i], (long long)appearance[i], (long long)rank[i],
                                    (long long)cracked[i], ratio[i]);
        if (!e) return -2;
        p = (unsigned char *)e;
    }
    return (int64_t)((char *)p - buf);
}

unsigned usable_threads() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}

}  // namespace

namespace {
// Runs `work` on up to `threads` threads, the caller's included.  A thread that cannot be started (EAGAIN under a
// container's thread limit) is simply not there: the others, and in the end the caller alone, take every block.
template <typename F>
void run_on_threads(unsigned threads, F &work) {
    std::vector<std::thread> pool;
    try {
        pool.reserve(threads);
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back([&work]() { work(); });
    } catch (...) {
    }
    work();
    for (auto &t : pool) t.join();
}
}  // namespace

extern "C" int64_t mc_format_gc_cps(const uint32_t *cps, const uint64_t *off, uint64_t n_lines, const uint32_t *order,
                                    uint64_t n, const double *mlp, const int64_t *appearance, const int64_t *rank,
                                    const int64_t *cracked, const double *ratio, char *dst, size_t cap) {
    if (n && (!cps || !off || !order || !mlp || !appearance || !rank || !cracked || !ratio || !dst)) {
        mc_set_error("mc_format_gc_cps: bad argument");
        return MC_ERR_ARG;
    }
    try {
        const uint64_t blocks = (n + FORMAT_BLOCK - 1) / FORMAT_BLOCK;
        std::vector<std::unique_ptr<char[]>> buf(blocks);
        std::vector<int64_t> size(blocks, 0);
        std::atomic<uint64_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&]() {
            for (uint64_t b = next.fetch_add(1); b < blocks && !bad.load(std::memory_order_relaxed); b = next.fetch_add(1)) {
                const uint64_t r0 = b * FORMAT_BLOCK, r1 = std::min(n, r0 + FORMAT_BLOCK);
                uint64_t chars = 0;
                for (uint64_t i = r0; i < r1; ++i) {
                    // a row names one of the n_lines unique lines, and the lines' offsets ascend: checked before
                    // anything is read through them
                    if (order[i] >= n_lines || off[order[i] + 1] < off[order[i]]) {
                        bad.store(4);
                        return;
                    }
                    chars += off[order[i] + 1] - off[order[i]];
                }
                buf[b].reset(new (std::nothrow) char[4 * chars + 120 * (r1 - r0)]);
                if (!buf[b]) {
                    bad.store(3);
                    return;
                }
                size[b] = format_rows_cps(cps, off, order, r0, r1, mlp, appearance, rank, cracked, ratio, buf[b].get());
                if (size[b] < 0) bad.store(size[b] == -1 ? 1 : 2);
            }
        };
        const unsigned threads = (unsigned)std::min<uint64_t>(usable_threads(), blocks);
        run_on_threads(threads, work);
        if (bad.load()) {
            const int why = bad.load();
            mc_set_error(why == 1   ? "mc_format_gc_cps: a lone surrogate has no UTF-8 form"
                         : why == 2 ? "mc_format_gc_cps: a line's numbers do not fit 120 characters"
                         : why == 4 ? "mc_format_gc_cps: a row names a line beyond the n_lines unique lines (or the "
                                      "offsets do not ascend)"
                                    : "mc_format_gc_cps: out of memory");
            return why == 3 ? MC_ERR_NOMEM : MC_ERR_ARG;
        }
        std::vector<uint64_t> at(blocks + 1, 0);
        for (uint64_t b = 0; b < blocks; ++b) at[b + 1] = at[b] + (uint64_t)size[b];
        if (at[blocks] > cap) {
            mc_set_error("mc_format_gc_cps: output buffer of %zu bytes is too small", cap);
            return MC_ERR_WORKSPACE;
        }
        next.store(0);
        auto place = [&]() {
            for (uint64_t b = next.fetch_add(1); b < blocks; b = next.fetch_add(1)) {
                std::memcpy(dst + at[b], buf[b].get(), (size_t)size[b]);
                buf[b].reset();
            }
        };
        run_on_threads(threads, place);
        return (int64_t)at[blocks];
    } catch (const std::bad_alloc &) {  // (nothing of C++ may leave an extern "C" function: the caller is ctypes)
        mc_set_error("mc_format_gc_cps: out of memory");
        return MC_ERR_NOMEM;
    } catch (...) {
        mc_set_error("mc_format_gc_cps: an unexpected C++ exception");
        return MC_ERR_ARG;
    }
}
