// split.cpp -- native pre-split for the two GPT split patterns.
//
// The reference chunks its input with `regex.findall(pattern, text)`
// (minbpe/regex.py:18-19, 41, 114).  The `regex` module runs these patterns at
// ~5 MB/s; on a 1 GB corpus that is minutes of host time in front of a training
// run that takes seconds on the GPU.  This file is a hand-written scanner for
// exactly those two patterns, over UTF-8 bytes, producing chunk START offsets.
// Character classes come from unicode_tables.h, generated from the `regex`
// module itself; tests/test_split.py checks equality with `regex.findall` on
// adversarial and random Unicode text.  Any other pattern stays with `regex`.
//
// GPT-4: '(?i:[sdmt]|ll|ve|re) | [^\r\n\p{L}\p{N}]?+\p{L}+ | \p{N}{1,3}
//        |  ?[^\s\p{L}\p{N}]++[\r\n]* | \s*[\r\n] | \s+(?!\S) | \s+
// GPT-2: '(?:[sdmt]|ll|ve|re) |  ?\p{L}+ |  ?\p{N}+ |  ?[^\s\p{L}\p{N}]+ | \s+(?!\S) | \s+
// Alternation is ordered: the first alternative that matches at a position wins.
// Every character is matched by some alternative, so the chunks partition the text.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "bpe_hip.h"
#include "unicode_tables.h"

namespace {

enum { C_OTHER = 0, C_L = 1, C_N = 2, C_S = 3 };

inline int cls(uint32_t cp) {
    if (cp < 256) return UC_STAGE2[UC_STAGE1[0]][cp];
    return cp < 0x110000 ? UC_STAGE2[UC_STAGE1[cp >> 8]][cp & 255] : C_OTHER;
}
// the classes of the ASCII bytes: most of any text, one load instead of decode + two-stage lookup
const uint8_t *const ASCII_CLS = UC_STAGE2[UC_STAGE1[0]];

// decode one code point of valid UTF-8 at p (p < e); len = its byte length
inline uint32_t dec(const uint8_t *p, const uint8_t *e, uint32_t &len) {
    const uint32_t b0 = p[0];
    if (b0 < 0x80) {
        len = 1;
        return b0;
    }
    if (b0 < 0xE0 && p + 1 < e) {
        len = 2;
        return ((b0 & 0x1F) << 6) | (p[1] & 0x3F);
    }
    if (b0 < 0xF0 && p + 2 < e) {
        len = 3;
        return ((b0 & 0x0F) << 12) | ((p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    }
    if (p + 3 < e) {
        len = 4;
        return ((b0 & 0x07) << 18) | ((p[1] & 0x3F) << 12) | ((p[2] & 0x3F) << 6) | (p[3] & 0x3F);
    }
    len = 1;  // truncated sequence: treat the byte as one "other" character
    return 0xFFFD;
}

template <size_t K>
inline bool in_set(uint32_t cp, const uint32_t (&set)[K]) {
    for (size_t i = 0; i < K; i++)
        if (set[i] == cp) return true;
    return false;
}

// length in bytes of the contraction suffix after an apostrophe at p (0: none)
template <bool ICASE>
inline uint32_t contraction(const uint8_t *p, const uint8_t *e) {
    if (p >= e) return 0;
    uint32_t l1, l2;
    const uint32_t c1 = dec(p, e, l1);
    if (ICASE) {
        if (in_set(c1, UC_FOLD_S) || in_set(c1, UC_FOLD_D) || in_set(c1, UC_FOLD_M) || in_set(c1, UC_FOLD_T))
            return l1;
        if (p + l1 >= e) return 0;
        const uint32_t c2 = dec(p + l1, e, l2);
        if (in_set(c1, UC_FOLD_L) && in_set(c2, UC_FOLD_L)) return l1 + l2;
        if (in_set(c1, UC_FOLD_V) && in_set(c2, UC_FOLD_E)) return l1 + l2;
        if (in_set(c1, UC_FOLD_R) && in_set(c2, UC_FOLD_E)) return l1 + l2;
        return 0;
    }
    if (c1 == 's' || c1 == 'd' || c1 == 'm' || c1 == 't') return 1;
    if (p + 1 >= e) return 0;
    const uint32_t c2 = p[1];
    if ((c1 == 'l' && c2 == 'l') || (c1 == 'v' && c2 == 'e') || (c1 == 'r' && c2 == 'e')) return 2;
    return 0;
}

// end of the run of characters of class `c` starting at p
inline const uint8_t *run_of(const uint8_t *p, const uint8_t *e, int c) {
    while (p < e) {
        if (*p < 0x80) {  // (ASCII fast path)
            if (ASCII_CLS[*p] != c) break;
            p++;
            continue;
        }
        uint32_t l;
        if (cls(dec(p, e, l)) != c) break;
        p += l;
    }
    return p;
}

struct Sink {
    std::vector<uint64_t> v;
    inline void put(uint64_t off) { v.push_back(off); }
};

// the sinks' offsets, one after the other, into `out` (every sink copied by its own thread when there is much to copy:
// 170 M offsets of a 1 GB text are 1.4 GB, a fifth of a second for one core)
void copy_out(const std::vector<Sink> &sinks, uint64_t *out, uint64_t total) {
    std::vector<uint64_t> at(sinks.size() + 1, 0);
    for (size_t i = 0; i < sinks.size(); i++) at[i + 1] = at[i] + sinks[i].v.size();
    auto one = [&](size_t i) {
        if (!sinks[i].v.empty()) memcpy(out + at[i], sinks[i].v.data(), sinks[i].v.size() * sizeof(uint64_t));
    };
    if (sinks.size() == 1 || total < (1u << 20)) {
        for (size_t i = 0; i < sinks.size(); i++) one(i);
        return;
    }
    std::vector<std::thread> th;
    for (size_t i = 0; i < sinks.size(); i++) th.emplace_back(one, i);
    for (auto &t : th) t.join();
}

// Chunks that START in [b, stop); matching looks ahead up to the end of the text `e` (a
// segment end is not the end of the text).  Starts are reported relative to `base`.
template <bool GPT4>
void scan(const uint8_t *base, const uint8_t *b, const uint8_t *stop, const uint8_t *e, Sink &sink) {
    const uint8_t *p = b;
    while (p < stop) {
        sink.put((uint64_t)(p - base));
        uint32_t l0 = 1;
        uint32_t c0 = *p;
        int k0;
        if (c0 < 0x80) {
            k0 = ASCII_CLS[c0];
        } else {
            c0 = dec(p, e, l0);
            k0 = cls(c0);
        }
        // alt 1: contractions
        if (c0 == '\'') {
            const uint32_t sl = contraction<GPT4>(p + 1, e);
            if (sl) {
                p += 1 + sl;
                continue;
            }
        }
        if (GPT4) {
            // alt 2: [^\r\n\p{L}\p{N}]?+\p{L}+   (possessive: the prefix is taken whenever it can be)
            const uint8_t *q = p;
            if (k0 != C_L && k0 != C_N && c0 != '\r' && c0 != '\n') q = p + l0;
            if (q < e) {
                uint32_t l;
                if (cls(dec(q, e, l)) == C_L) {
                    p = run_of(q + l, e, C_L);
                    continue;
                }
            }
            // alt 3: \p{N}{1,3}
            if (k0 == C_N) {
                const uint8_t *r = p + l0;
                for (int i = 1; i < 3 && r < e; i++) {
                    uint32_t l;
                    if (cls(dec(r, e, l)) != C_N) break;
                    r += l;
                }
                p = r;
                continue;
            }
            // alt 4:  ?[^\s\p{L}\p{N}]++[\r\n]*
            q = (c0 == ' ') ? p + 1 : p;
            if (q < e) {
                uint32_t l;
                if (cls(dec(q, e, l)) == C_OTHER) {
                    const uint8_t *r = run_of(q + l, e, C_OTHER);
                    while (r < e && (*r == '\r' || *r == '\n')) r++;
                    p = r;
                    continue;
                }
            }
        } else {
            // GPT-2 alt 2-4:  ?\p{L}+ |  ?\p{N}+ |  ?[^\s\p{L}\p{N}]+
            const uint8_t *q = (c0 == ' ') ? p + 1 : p;
            if (q < e) {
                uint32_t l;
                const int kq = cls(dec(q, e, l));
                if (kq != C_S) {
                    p = run_of(q + l, e, kq);
                    continue;
                }
            }
        }
        // whitespace alternatives (k0 == C_S from here on; anything else matched above)
        if (k0 == C_S) {
            const uint8_t *r = p + l0;  // end of the whitespace run
            const uint8_t *last_nl = (c0 == '\r' || c0 == '\n') ? p + l0 : nullptr;  // end of the last \r|\n
            const uint8_t *last_ch = p;                                              // start of the run's last char
            while (r < e) {
                uint32_t l;
                const uint32_t c = dec(r, e, l);
                if (cls(c) != C_S) break;
                last_ch = r;
                r += l;
                if (c == '\r' || c == '\n') last_nl = r;
            }
            if (GPT4 && last_nl) {  // \s*[\r\n]: up to the last newline of the run
                p = last_nl;
                continue;
            }
            if (r == e) {  // \s+(?!\S) at the end of the text: the whole run
                p = r;
                continue;
            }
            if (last_ch > p) {  // \s+(?!\S): the run minus its last character
                p = last_ch;
                continue;
            }
            p = r;  // \s+: a single whitespace character before a non-space
            continue;
        }
        p += l0;  // unreachable for valid input: every class is matched above
    }
}

template <bool GPT4>
int split_impl(const uint8_t *s, uint64_t n, uint64_t *out, uint64_t cap, uint64_t *n_chunks, int threads) {
    const uint8_t *e = s + n;
    // Segments start at "safe" boundaries: right after a '\n' that is followed by a
    // non-whitespace ASCII byte or a non-whitespace character -- every alternative ends a
    // chunk there (newline-terminated punctuation, \s*[\r\n], and no chunk starts with \n
    // and continues into a letter), so the segments can be scanned independently.
    std::vector<const uint8_t *> cuts{s};
    if (threads > 1 && n > (1u << 20)) {
        for (int t = 1; t < threads; t++) {
            const uint8_t *p = s + n / threads * t;
            const uint8_t *lim = std::min(e, p + (1u << 20));
            const uint8_t *found = nullptr;
            for (; p + 1 < lim; p++) {
                if (p[0] == '\n' && p[1] < 0x80 && p[1] > ' ' && p[1] != 0x7f) {  // printable ASCII after \n
                    found = p + 1;
                    break;
                }
            }
            if (found && found > cuts.back()) cuts.push_back(found);
        }
    }
    cuts.push_back(e);
    const size_t nseg = cuts.size() - 1;
    std::vector<Sink> sinks(nseg);
    auto scan_seg = [&](size_t i) {
        sinks[i].v.reserve((size_t)(cuts[i + 1] - cuts[i]) / 4 + 16);
        scan<GPT4>(s, cuts[i], cuts[i + 1], e, sinks[i]);
    };
    if (nseg == 1) {
        scan_seg(0);
    } else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < nseg; i++) th.emplace_back(scan_seg, i);
        for (auto &t : th) t.join();
    }
    uint64_t total = 0;
    for (auto &sk : sinks) total += sk.v.size();
    if (n_chunks) *n_chunks = total;
    if (!out) return BPE_OK;
    if (cap < total) return BPE_E_CAP;
    copy_out(sinks, out, total);
    return BPE_OK;
}

// documents: every document is split on its own (a match never crosses a document boundary)
template <bool GPT4>
int split_docs_impl(const uint8_t *s, uint64_t n, const uint64_t *doc_off, uint64_t n_docs, uint64_t *out,
                    uint64_t cap, uint64_t *n_chunks, uint64_t *doc_first_chunk, int threads) {
    auto doc_begin = [&](uint64_t d) { return doc_off[d]; };
    auto doc_end = [&](uint64_t d) { return d + 1 < n_docs ? doc_off[d + 1] : n; };
    // contiguous groups of documents with about the same number of bytes
    const unsigned T = (unsigned)std::max(1, threads);
    std::vector<uint64_t> cut(T + 1, n_docs);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) {
        const uint64_t target = n / T * t;
        cut[t] = (uint64_t)(std::lower_bound(doc_off, doc_off + n_docs, target) - doc_off);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<Sink> sinks(T);
    std::vector<uint64_t> per_doc(n_docs, 0);
    auto work = [&](unsigned t) {
        Sink &sk = sinks[t];
        for (uint64_t d = cut[t]; d < cut[t + 1]; d++) {
            const size_t before = sk.v.size();
            const uint64_t b = doc_begin(d), e = doc_end(d);
            if (e > b) scan<GPT4>(s, s + b, s + e, s + e, sk);
            per_doc[d] = sk.v.size() - before;
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    uint64_t total = 0;
    for (uint64_t d = 0; d < n_docs; d++) {
        if (doc_first_chunk) doc_first_chunk[d] = total;
        total += per_doc[d];
    }
    if (doc_first_chunk) doc_first_chunk[n_docs] = total;
    if (n_chunks) *n_chunks = total;
    if (!out) return BPE_OK;
    if (cap < total) return BPE_E_CAP;
    copy_out(sinks, out, total);
    return BPE_OK;
}

// one thread scans a few hundred MB/s; threads only pay on long texts: one per 8 MB, at most 112 (measured on a
// 256-thread host, 1 GB: 0.25 s with 32 threads, 0.18 with 64, 0.11 with 96..128, 0.13 with 192)
int auto_threads(uint64_t n) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min(112u, hw), n >> 23));
}

}  // namespace

extern "C" int bpe_split_docs(int which, const uint8_t *utf8, uint64_t n, const uint64_t *doc_offsets,
                              uint64_t n_docs, uint64_t *starts_out, uint64_t cap, uint64_t *n_chunks,
                              uint64_t *doc_first_chunk, int threads) {
    if ((!utf8 && n) || (which != 2 && which != 4) || (!doc_offsets && n_docs)) return BPE_E_ARG;
    for (uint64_t d = 0; d < n_docs; d++) {
        const uint64_t b = doc_offsets[d], e = d + 1 < n_docs ? doc_offsets[d + 1] : n;
        if (b > e || e > n) return BPE_E_ARG;  // offsets must ascend and stay inside the text
    }
    if (threads < 1) threads = auto_threads(n);
    if (n_docs == 0) {
        if (n_chunks) *n_chunks = 0;
        if (doc_first_chunk) doc_first_chunk[0] = 0;
        return BPE_OK;
    }
    return which == 4 ? split_docs_impl<true>(utf8, n, doc_offsets, n_docs, starts_out, cap, n_chunks,
                                              doc_first_chunk, threads)
                      : split_docs_impl<false>(utf8, n, doc_offsets, n_docs, starts_out, cap, n_chunks,
                                               doc_first_chunk, threads);
}

extern "C" int bpe_split(int which, const uint8_t *utf8, uint64_t n, uint64_t *starts_out, uint64_t cap,
                         uint64_t *n_chunks, int threads) {
    if ((!utf8 && n) || (which != 2 && which != 4)) return BPE_E_ARG;
    if (threads < 1) threads = auto_threads(n);
    if (n == 0) {
        if (n_chunks) *n_chunks = 0;
        return BPE_OK;
    }
    return which == 4 ? split_impl<true>(utf8, n, starts_out, cap, n_chunks, threads)
                      : split_impl<false>(utf8, n, starts_out, cap, n_chunks, threads);
}
