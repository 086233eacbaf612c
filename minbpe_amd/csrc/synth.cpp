// synth.cpp -- deterministic synthetic UTF-8 text (SURVEY.md section 8d,
// "synth_text").  Host-only utility exported through the C-ABI so that the
// benchmark, the tests and the CPU oracle all see byte-identical inputs.
//
// Explicit splitmix64; every random choice is an integer threshold compare, so
// the stream does not depend on numpy / Python versions.  tests/ pin the sha256
// of the first MiB.
//
// Model: Zipf(1.1) over a 65,536-word lexicon; word length 1+Poisson(4.5)
// clipped to 1..16 characters; 94 % ASCII-letter words (English letter
// frequencies), 5 % two-byte (Latin-1 / Greek / Cyrillic), 0.8 % three-byte CJK,
// 0.2 % four-byte emoji; 2 % of tokens are 1-6 digit numbers; separators
// ' ' 85 %, ', ' 5 %, '. ' 5 %, '\n' 4 %, '\n\n' 1 %.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "bpe_hip.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t m) { return (uint32_t)((next() >> 32) * (uint64_t)m >> 32); }
};

void put_utf8(std::string &o, uint32_t cp) {
    if (cp < 0x80) {
        o.push_back((char)cp);
    } else if (cp < 0x800) {
        o.push_back((char)(0xC0 | (cp >> 6)));
        o.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        o.push_back((char)(0xE0 | (cp >> 12)));
        o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        o.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        o.push_back((char)(0xF0 | (cp >> 18)));
        o.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        o.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

constexpr int LEX = 65536;

struct Lexicon {
    std::vector<std::string> words;
    std::vector<uint64_t> zipf_cdf;  // thresholds on a 2^53 scale
    Lexicon() {
        // letter frequencies (per 10000) for a..z
        static const int freq[26] = {817, 149, 278, 425, 1270, 223, 202, 609, 697, 15,  77,  403, 241,
                                     675, 751, 193, 10,  599,  633, 906, 276, 98,  236, 15,  197, 7};
        uint32_t lcdf[26];
        uint32_t acc = 0;
        for (int i = 0; i < 26; i++) {
            acc += (uint32_t)freq[i];
            lcdf[i] = acc;
        }
        // 1 + Poisson(4.5), clipped to 1..16: thresholds on a 2^32 scale
        uint64_t pcdf[16];
        {
            double p = exp(-4.5), c = 0.0;
            for (int k = 0; k < 16; k++) {
                c += p;
                pcdf[k] = (k == 15) ? 0xFFFFFFFFull : (uint64_t)(c * 4294967296.0);
                p *= 4.5 / (double)(k + 1);
            }
        }
        words.resize(LEX);
        for (int w = 0; w < LEX; w++) {
            Rng r(0xC0FFEE1234ULL ^ ((uint64_t)w * 0x9E3779B97F4A7C15ULL));
            const uint32_t cls = r.below(1000);
            const uint64_t u = r.next() >> 32;
            int len = 1;
            while (len < 16 && u >= pcdf[len - 1]) len++;
            std::string &s = words[w];
            if (cls < 940) {
                const bool cap = r.below(100) < 5;
                for (int i = 0; i < len; i++) {
                    const uint32_t x = r.below(acc);
                    int c = 0;
                    while (x >= lcdf[c]) c++;
                    s.push_back((char)((i == 0 && cap ? 'A' : 'a') + c));
                }
            } else if (cls < 990) {
                const uint32_t script = r.below(3);
                for (int i = 0; i < len; i++) {
                    uint32_t cp;
                    if (script == 0) {
                        cp = 0xC0 + r.below(0x40);
                        if (cp == 0xD7 || cp == 0xF7) cp = 0xE9;
                    } else if (script == 1) {
                        cp = 0x3B1 + r.below(25);
                        if (cp == 0x3C2) cp = 0x3C3;
                    } else {
                        cp = 0x430 + r.below(32);
                    }
                    put_utf8(s, cp);
                }
            } else if (cls < 998) {
                const int l3 = len > 4 ? 4 : len;
                for (int i = 0; i < l3; i++) put_utf8(s, 0x4E00 + r.below(0x5000));
            } else {
                const int l4 = len > 2 ? 2 : 1;
                for (int i = 0; i < l4; i++) put_utf8(s, 0x1F600 + r.below(0x50));
            }
        }
        zipf_cdf.resize(LEX);
        double tot = 0.0;
        for (int k = 1; k <= LEX; k++) tot += pow((double)k, -1.1);
        double c = 0.0;
        for (int k = 1; k <= LEX; k++) {
            c += pow((double)k, -1.1);
            zipf_cdf[k - 1] = (k == LEX) ? (1ull << 53) : (uint64_t)(c / tot * 9007199254740992.0);
        }
    }
};

const Lexicon &lexicon() {
    static Lexicon L;
    return L;
}

}  // namespace

extern "C" int bpe_synth_text(uint8_t *out, uint64_t n, uint64_t seed) {
    if (!out && n) return BPE_E_ARG;
    const Lexicon &L = lexicon();
    Rng r(seed * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL);
    uint64_t w = 0;
    char num[8];
    while (w < n) {
        const char *tok;
        size_t tl;
        if (r.below(100) < 2) {
            const int nd = 1 + (int)r.below(6);
            for (int i = 0; i < nd; i++) num[i] = (char)('0' + r.below(10));
            tok = num;
            tl = (size_t)nd;
        } else {
            const uint64_t u = r.next() >> 11;  // 53 bits
            size_t lo = 0, hi = LEX - 1;
            while (lo < hi) {
                const size_t mid = (lo + hi) >> 1;
                if (u < L.zipf_cdf[mid]) hi = mid; else lo = mid + 1;
            }
            tok = L.words[lo].data();
            tl = L.words[lo].size();
        }
        const uint32_t sp = r.below(100);
        const char *sep = sp < 85 ? " " : sp < 90 ? ", " : sp < 95 ? ". " : sp < 99 ? "\n" : "\n\n";
        const size_t sl = strlen(sep);
        if (w + tl + sl > n) {  // never cut inside a character: pad with spaces
            memset(out + w, ' ', n - w);
            w = n;
            break;
        }
        memcpy(out + w, tok, tl);
        memcpy(out + w + tl, sep, sl);
        w += tl + sl;
    }
    return BPE_OK;
}
