// utf8.cpp -- text.encode("utf-8") by all host threads (host only, no GPU).
//
// Every train() / encode() of the reference starts with `text.encode("utf-8")` (basic.py:25, regex.py:44): CPython does it
// with one thread, 0.7 s per GB of a str that holds characters beyond the BMP -- more than the whole device side of a
// train() to vocab 32000 takes.  A CPython str is an array of code points of 1, 2 or 4 bytes each; this file turns such
// an array into UTF-8 in two segment-parallel passes (count, then write).  Lone surrogates and values above 0x10FFFF are
// refused (the caller falls back to str.encode, which raises what the reference would raise).
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "bpe_hip.h"

namespace {

template <typename T>
inline bool count_seg(const T *p, uint64_t n, uint64_t &bytes) {
    uint64_t b = 0;
    bool ok = true;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = (uint32_t)p[i];
        b += 1 + (c >= 0x80) + (c >= 0x800) + (c >= 0x10000);
        if (sizeof(T) > 1) ok &= !((c >= 0xD800 && c < 0xE000) || c > 0x10FFFF);
    }
    bytes = b;
    return ok;
}
template <typename T>
inline void write_seg(const T *p, uint64_t n, uint8_t *o) {
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = (uint32_t)p[i];
        if (c < 0x80) {
            *o++ = (uint8_t)c;
        } else if (c < 0x800) {
            *o++ = (uint8_t)(0xC0 | (c >> 6));
            *o++ = (uint8_t)(0x80 | (c & 0x3F));
        } else if (c < 0x10000) {
            *o++ = (uint8_t)(0xE0 | (c >> 12));
            *o++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F));
            *o++ = (uint8_t)(0x80 | (c & 0x3F));
        } else {
            *o++ = (uint8_t)(0xF0 | (c >> 18));
            *o++ = (uint8_t)(0x80 | ((c >> 12) & 0x3F));
            *o++ = (uint8_t)(0x80 | ((c >> 6) & 0x3F));
            *o++ = (uint8_t)(0x80 | (c & 0x3F));
        }
    }
}

template <typename T>
int encode_impl(const T *cps, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *n_bytes, int threads) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned T_ = threads > 0 ? (unsigned)threads : (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(32u, hw), n >> 22));  // (bound by memory: 0.06 s per GB with 32 threads, 0.10 with 64, 0.18 with 128)
    T_ = std::max(1u, T_);
    std::vector<uint64_t> cut(T_ + 1), bytes(T_, 0);
    std::vector<char> ok(T_, 1);
    for (unsigned t = 0; t <= T_; t++) cut[t] = n / T_ * t;
    cut[T_] = n;
    auto par = [&](auto fn) {
        if (T_ == 1) {
            fn(0u);
            return;
        }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T_; t++) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    par([&](unsigned t) { ok[t] = count_seg(cps + cut[t], cut[t + 1] - cut[t], bytes[t]) ? 1 : 0; });
    uint64_t total = 0;
    for (unsigned t = 0; t < T_; t++) {
        if (!ok[t]) return BPE_E_ARG;
        const uint64_t b = bytes[t];
        bytes[t] = total;  // (now: where the segment's bytes start)
        total += b;
    }
    if (n_bytes) *n_bytes = total;
    if (!out) return BPE_OK;
    if (cap < total) return BPE_E_CAP;
    par([&](unsigned t) { write_seg(cps + cut[t], cut[t + 1] - cut[t], out + bytes[t]); });
    return BPE_OK;
}

}  // namespace

extern "C" int bpe_utf8_encode(int kind, const void *code_points, uint64_t n_chars, uint8_t *out, uint64_t cap,
                               uint64_t *n_bytes, int threads) {
    if ((!code_points && n_chars) || (kind != 1 && kind != 2 && kind != 4)) return BPE_E_ARG;
    if (kind == 1) return encode_impl((const uint8_t *)code_points, n_chars, out, cap, n_bytes, threads);
    if (kind == 2) return encode_impl((const uint16_t *)code_points, n_chars, out, cap, n_bytes, threads);
    return encode_impl((const uint32_t *)code_points, n_chars, out, cap, n_bytes, threads);
}
