// dedup.cpp -- chunk de-duplication for chunked training.
//
// The reference keeps every regex chunk of the text (regex.py:41-44) and counts pairs over
// all of them (regex.py:51-54).  Natural text repeats its chunks heavily ("the", " of",
// ...): counting a distinct chunk once with a weight gives the same statistics, and -- when
// the distinct chunks are kept in order of FIRST APPEARANCE -- the same dict insertion
// order, hence the same max() tie-breaks: the first occurrence of any pair lies in the
// first instance of some chunk, and identical chunks stay identical under every merge.
// The device counts with power-of-two weights (bpe_device.h), so a chunk of multiplicity w
// is emitted once per set bit of w.
//
// Host side and exact.  Each thread counts a contiguous range of chunks in its own
// open-addressing table (32-byte entries: one cache line per probe; a chunk of up to 8
// bytes IS its key, longer ones are hashed and confirmed with memcmp), the per-thread
// tables are merged pairwise, the later range into the earlier one (so "first" stays the
// first appearance), and the result is sorted by first appearance.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "bpe_hip.h"

namespace {

inline uint64_t mix(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}

inline uint64_t hash_long(const uint8_t *p, uint64_t len) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (len * 0xff51afd7ed558ccdull);
    while (len >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = mix(h ^ w) + 0x2545F4914F6CDD1Dull;
        p += 8;
        len -= 8;
    }
    if (len) {
        uint64_t w = 0;
        memcpy(&w, p, len);
        h = mix(h ^ w);
    }
    return h;
}

struct alignas(32) Entry {
    uint64_t key;    // len <= 8: the bytes themselves, zero-padded; else a 64-bit hash
    uint64_t first;  // index of the first chunk with this content
    uint64_t count;  // 0: free slot
    uint64_t len;
};

struct Span {
    const uint8_t *bytes;
    const uint64_t *off;
    uint64_t n, n_chunks;
    inline uint64_t begin(uint64_t c) const { return off[c]; }
    inline uint64_t end(uint64_t c) const { return c + 1 < n_chunks ? off[c + 1] : n; }
};

struct Table {
    std::vector<Entry> slot;
    uint64_t mask = 0, used = 0;
    const Span *sp = nullptr;

    void init(const Span *s, uint64_t cap) {
        sp = s;
        uint64_t c = 1024;
        while (c < cap) c <<= 1;
        slot.assign(c, Entry{0, 0, 0, 0});
        mask = c - 1;
        used = 0;
    }
    inline bool same(const Entry &e, uint64_t key, uint64_t len, const uint8_t *p) const {
        if (e.key != key || e.len != len) return false;
        return len <= 8 || memcmp(sp->bytes + sp->begin(e.first), p, len) == 0;
    }
    // add `count` occurrences of the chunk (p, len) whose first appearance is chunk `first`
    inline void add(uint64_t key, uint64_t len, const uint8_t *p, uint64_t first, uint64_t count) {
        uint64_t h = mix(key ^ (len << 56)) & mask;
        for (;;) {
            Entry &e = slot[h];
            if (!e.count) {
                e = Entry{key, first, count, len};
                if (++used * 2 > mask) grow();
                return;
            }
            if (same(e, key, len, p)) {
                e.count += count;
                return;
            }
            h = (h + 1) & mask;
        }
    }
    void grow() {
        std::vector<Entry> old;
        old.swap(slot);
        slot.assign(old.size() * 2, Entry{0, 0, 0, 0});
        mask = slot.size() - 1;
        for (const Entry &e : old) {
            if (!e.count) continue;
            uint64_t h = mix(e.key ^ (e.len << 56)) & mask;
            while (slot[h].count) h = (h + 1) & mask;
            slot[h] = e;
        }
    }
};

inline uint64_t key_of(const uint8_t *p, uint64_t len) {
    if (len > 8) return hash_long(p, len);
    uint64_t w = 0;
    memcpy(&w, p, len);
    return w;
}

}  // namespace

extern "C" int bpe_dedup_chunks(const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                                uint64_t n_chunks, uint8_t *out_bytes, uint64_t *out_offsets,
                                uint8_t *out_weight_exp, uint64_t *n_out_bytes, uint64_t *n_out_chunks,
                                uint64_t *n_distinct, int threads) {
    if ((!bytes && n) || (!chunk_offsets && n_chunks) || !out_offsets || !out_weight_exp || (!out_bytes && n))
        return BPE_E_ARG;
    if (n_chunks >= (1ull << 32)) return BPE_E_LIMIT;  // multiplicities must fit the 32 weight exponents
    if (threads < 1) {  // one per million chunks, at most 64
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        threads = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min(64u, hw), n_chunks >> 20));
    }
    if (n_chunks < (1u << 16)) threads = 1;
    const Span sp{bytes, chunk_offsets, n, n_chunks};
    const unsigned T = (unsigned)threads;

    // 1. one table per contiguous range of chunks
    //    (every range first checks its own offsets: they must ascend and stay inside the text)
    std::vector<Table> local(T);
    std::vector<uint8_t> bad(T, 0);
    auto count_range = [&](unsigned t) {
        Table &tb = local[t];
        tb.init(&sp, 1 << 14);
        const uint64_t c0 = n_chunks * t / T, c1 = n_chunks * (t + 1) / T;
        for (uint64_t c = c0; c < c1; c++) {
            const uint64_t b = sp.begin(c), e = sp.end(c);
            if (b > e || e > n) {
                bad[t] = 1;
                return;
            }
        }
        for (uint64_t c = c0; c < c1; c++) {
            const uint64_t b = sp.begin(c), len = sp.end(c) - b;
            tb.add(key_of(bytes + b, len), len, bytes + b, c, 1);
        }
    };
    if (T == 1) {
        count_range(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back(count_range, t);
        for (auto &x : th) x.join();
    }
    for (unsigned t = 0; t < T; t++)
        if (bad[t]) return BPE_E_ARG;
    // 2. merge, always the later range INTO the earlier one (the earlier range's first appearance stands), as a tree:
    //    round r folds table t + 2^r into table t for every t that is a multiple of 2^(r+1), all folds of a round at
    //    once -- log2(T) rounds of one table's worth of inserts each instead of T tables through one thread (with 64
    //    ranges of a text whose distinct chunks number 10^5 the serial merge was most of the pass)
    auto fold = [&](unsigned dst, unsigned src) {
        for (const Entry &e : local[src].slot)
            if (e.count) local[dst].add(e.key, e.len, bytes + sp.begin(e.first), e.first, e.count);
        std::vector<Entry>().swap(local[src].slot);
    };
    for (unsigned step = 1; step < T; step <<= 1) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t + step < T; t += 2 * step) th.emplace_back(fold, t, t + step);
        for (auto &x : th) x.join();
    }
    Table *all = &local[0];
    // 3. distinct chunks by first appearance, one copy per set bit of the multiplicity
    std::vector<std::pair<uint64_t, uint64_t>> order;  // (first, count)
    order.reserve(all->used);
    for (const Entry &e : all->slot)
        if (e.count) order.emplace_back(e.first, e.count);
    std::sort(order.begin(), order.end());
    uint64_t wb = 0, wc = 0;
    for (auto &fc : order) {
        const uint64_t b = sp.begin(fc.first), len = sp.end(fc.first) - b;
        for (uint64_t k = 0, w = fc.second; w; k++, w >>= 1) {
            if (!(w & 1)) continue;
            if (len) memcpy(out_bytes + wb, bytes + b, len);
            out_offsets[wc] = wb;
            out_weight_exp[wc] = (uint8_t)k;
            wb += len;
            wc++;
        }
    }
    if (n_out_bytes) *n_out_bytes = wb;
    if (n_out_chunks) *n_out_chunks = wc;
    if (n_distinct) *n_distinct = order.size();
    return BPE_OK;
}
