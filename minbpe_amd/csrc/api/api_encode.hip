// api_encode.hip -- bpe_encode_batch.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// ---------------------------------------------------------------------------
// encode

namespace {
inline uint64_t mix_key(uint64_t key) { return (key * 0x9E3779B97F4A7C15ull) >> 40; }

int upload_offsets(bpe_ctx *c, const uint64_t *chunk_offsets, uint64_t n_chunks) {
    if (n_chunks > c->cap_offsets) {
        TRY(dev_realloc(c, c->d_offsets, (size_t)n_chunks));
        c->cap_offsets = n_chunks;
    }
    if (n_chunks) TRY(upload_h2d(c, c->d_offsets, chunk_offsets, n_chunks * sizeof(uint64_t)));
    return BPE_OK;
}
}  // namespace

// 16-bit token / rank columns in k_encode_short: every rank (0xFFFF = "none") and every token id must fit.
// Without merge_ids token r is 256 + r: the last one, 255 + M, must still be below 65536.
extern "C" int bpe_encode_uses_16bit(const int32_t *merge_ids, int32_t M) {
    bool narrow = merge_ids ? M < 65535 : M <= 65536 - 256;
    for (int32_t r = 0; narrow && merge_ids && r < M; r++) narrow = merge_ids[r] >= 0 && merge_ids[r] < 65536;
    return narrow ? 1 : 0;
}

// bpe_encode_batch / bpe_encode_batch_resident.  resident: bytes, chunk_offsets, ids_out and out_offsets are DEVICE
// pointers (the rank table stays a host array: it is small) -- the kernels read the caller's offsets and write the
// caller's outputs in place, nothing crosses PCIe but the rank table and a few counters.
namespace {
__global__ void k_check_offsets(const uint64_t *__restrict__ off, uint64_t n_chunks, uint64_t n, unsigned long long *bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_chunks; i += stride)
        if (off[i] > n || (i && off[i] < off[i - 1])) atomicMin(bad, (unsigned long long)i);
}
__global__ void k_store_u64(uint64_t *p, unsigned long long v) { *p = v; }
int encode_batch_impl(bpe_ctx *c, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                      const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                      uint64_t n_chunks, int32_t *ids_out, uint64_t *out_offsets,
                      uint64_t *n_out, bool resident) {
    if (!c || M < 0 || (!merges && M) || (!bytes && n)) return fail(c, BPE_E_ARG, "bad arguments");
    if (n >= (1ull << 32)) return fail(c, BPE_E_LIMIT, "batch of %llu bytes exceeds 2^32-1", (unsigned long long)n);
    if (resident && (!chunk_offsets || !ids_out || !out_offsets))
        return fail(c, BPE_E_ARG, "the resident form needs chunk_offsets, ids_out and out_offsets on the device");
    if (!resident) TRY(check_offsets(c, chunk_offsets, n_chunks, n));
    static const uint64_t zero = 0;
    if (!chunk_offsets) {
        chunk_offsets = &zero;
        n_chunks = 1;
    }
    if (n_out) *n_out = 0;
    if (n == 0 || n_chunks == 0) {
        if (out_offsets && !resident)
            for (uint64_t i = 0; i <= n_chunks; i++) out_offsets[i] = 0;
        if (out_offsets && resident) {
            HIPCHK(c, hipSetDevice(c->device));
            HIPCHK(c, hipMemsetAsync(out_offsets, 0, (n_chunks + 1) * 8, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        return BPE_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (resident) {  // (the host form validates its offsets on the host)
        HIPCHK(c, hipMemsetAsync(c->d_scratch, 0xFF, 8, c->stream));
        hipLaunchKernelGGL(k_check_offsets, dim3(grid_for(n_chunks, 256, c->num_cus * 8)), dim3(256), 0, c->stream, chunk_offsets,
                           n_chunks, n, c->d_scratch);
        unsigned long long bad = 0;
        HIPCHK(c, hipMemcpyAsync(&bad, c->d_scratch, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (bad != ~0ull)
            return fail(c, BPE_E_ARG, "chunk_offsets[%llu]: offsets must ascend and stay <= n = %llu", bad, (unsigned long long)n);
    }
    // ---- ONE giant chunk (BasicTokenizer.encode: the whole text, basic.py:57-74) with a merge list of the shape training
    // makes -- consecutive ids, no pair twice, every pair made of tokens defined before it: replay the list through the
    // training engine (k_chain.hip: k_forced_sel / k_forced_pair).  The stream-wide rounds below sweep the whole stream once
    // per rank present (~3,000 sweeps and 6,000 host synchronisations for 100 MB at vocab 4096: 1.8 s); the replay is a
    // train() of the same text: slots, index, sparse sweeps, batches -- 0.1 s.
    constexpr uint64_t REPLAY_MIN_BYTES = 1u << 20;
    if (!resident && n_chunks == 1 && chunk_offsets[0] == 0 && !merge_ids && c->enc_replay && M > 0 && n >= REPLAY_MIN_BYTES &&
        256 + (int64_t)M <= 65535 && !c->dp_active) {
        bool shape = true;
        {
            std::vector<unsigned long long> seen;
            seen.reserve((size_t)M);
            for (int32_t r = 0; r < M && shape; r++) {
                const int32_t a = merges[2 * r], b = merges[2 * r + 1];
                shape = a >= 0 && b >= 0 && a < 256 + r && b < 256 + r;
                seen.push_back(((unsigned long long)(uint32_t)a << 32) | (uint32_t)b);
            }
            if (shape) {
                std::sort(seen.begin(), seen.end());
                shape = std::adjacent_find(seen.begin(), seen.end()) == seen.end();
            }
        }
        if (shape) {
            const int saved_mode = c->mode;
            c->mode = 1;
            int rc = bpe_load_bytes(c, bytes, n, nullptr, 0);
            if (rc == BPE_OK && (uint64_t)M > c->cap_forced) {
                rc = dev_realloc(c, c->d_forced, (size_t)2 * (size_t)M);
                if (rc == BPE_OK) c->cap_forced = (uint64_t)M;
            }
            if (rc == BPE_OK && hipMemcpyAsync(c->d_forced, merges, (size_t)2 * (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess)
                rc = fail(c, BPE_E_HIP, "upload of the merge list failed");
            int32_t done = 0;
            if (rc == BPE_OK) {
                c->forced = true;
                rc = bpe_train(c, M, nullptr, nullptr, nullptr, nullptr, &done);
                c->forced = false;
            }
            c->mode = saved_mode;
            if (rc == BPE_OK && done != M) rc = fail(c, BPE_E_INTERNAL, "replay stopped after %d of %d merges", done, M);
            if (rc != BPE_OK) return rc;
            const uint64_t total = c->n;
            if (ids_out && total) {
                int32_t *tmp = (int32_t *)c->d_ids[c->par ^ 1];
                hipLaunchKernelGGL(k_strip_flags, dim3(grid_for(total, 256, c->num_cus * 8)), dim3(256), 0, c->stream, c->d_ids[c->par], tmp, total);
                LAUNCHCHK(c, "k_strip_flags");
                TRY(download_d2h(c, ids_out, tmp, total * sizeof(int32_t)));
            }
            if (out_offsets) {
                out_offsets[0] = 0;
                out_offsets[1] = total;
            }
            if (n_out) *n_out = total;
            c->have_bytes = false;  // (this call reused the ctx's input and id-stream buffers)
            c->have_ids = false;
            c->stats_valid = false;
            TRY(prof_drain(c));
            return BPE_OK;
        }
    }
    // this call reuses the ctx's input and id-stream buffers
    c->have_bytes = false;
    c->weighted = false;
    c->have_ids = false;
    c->stats_valid = false;

    // 1. rank table: pair -> position in the (priority-ordered) merge list
    uint64_t hs = 16;
    while (hs < 2 * (uint64_t)M + 2) hs <<= 1;
    std::vector<unsigned long long> hk(hs, ~0ull);
    std::vector<uint32_t> hv(hs, 0xFFFFFFFFu);
    for (int32_t r = 0; r < M; r++) {
        const int32_t a = merges[2 * r], b = merges[2 * r + 1];
        if (a < 0 || b < 0) return fail(c, BPE_E_ARG, "negative id in merges[%d]", r);
        const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)b;
        uint64_t h = mix_key(key) & (hs - 1);
        while (hk[h] != ~0ull && hk[h] != key) h = (h + 1) & (hs - 1);
        hk[h] = key;
        hv[h] = (uint32_t)r;  // a repeated pair keeps its last entry, like a dict
    }
    if (hs > c->cap_ht) {
        TRY(dev_realloc(c, c->d_ht_keys, (size_t)hs));
        TRY(dev_realloc(c, c->d_ht_vals, (size_t)hs));
        c->cap_ht = hs;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ht_keys, hk.data(), hs * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ht_vals, hv.data(), hs * 4, hipMemcpyHostToDevice, c->stream));
    const int32_t *d_mids = nullptr;
    if (merge_ids && M) {
        if ((uint64_t)M > c->cap_merge_ids) {
            TRY(dev_realloc(c, c->d_merge_ids, (size_t)M));
            c->cap_merge_ids = (uint64_t)M;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_merge_ids, merge_ids, (size_t)M * 4, hipMemcpyHostToDevice, c->stream));
        d_mids = c->d_merge_ids;
    }
    // 2. input
    if (n + 16 > c->cap_bytes) {
        TRY(dev_realloc(c, c->d_bytes, (size_t)n + 16));
        c->cap_bytes = n + 16;
    }
    // (the bytes are copied either way: the kernels read up to 16 bytes past the end of the ctx's padded buffer)
    if (resident) HIPCHK(c, hipMemcpyAsync(c->d_bytes, bytes, n, hipMemcpyDeviceToDevice, c->stream));
    else TRY(upload_h2d(c, c->d_bytes, bytes, n));  // (pinned staging ring, several host threads: api_ctx.hip)
    if (!resident) TRY(upload_offsets(c, chunk_offsets, n_chunks));
    const uint64_t *d_offs = resident ? chunk_offsets : c->d_offsets;
    // 3. scratch
    if (n > c->cap_enc_n) {
        TRY(dev_realloc(c, c->d_enc_tmp, (size_t)n));
        TRY(dev_realloc(c, c->d_enc_out, (size_t)n));
        TRY(dev_realloc(c, c->d_enc_long, (size_t)(n / (ENC_LMAX + 1) + 2)));
        TRY(dev_realloc(c, c->d_enc_huge, (size_t)(n / (ENC_LONG_MAX + 1) + 2)));
        c->cap_enc_n = n;
    }
    const uint64_t nb = (n_chunks + SCAN_TILE - 1) / SCAN_TILE;
    if (n_chunks > c->cap_enc_chunks) {
        TRY(dev_realloc(c, c->d_enc_len, (size_t)n_chunks));
        TRY(dev_realloc(c, c->d_enc_off, (size_t)n_chunks + 1));
        TRY(dev_realloc(c, c->d_enc_bsum, (size_t)nb * (SCAN_TILE / ENC_PLACE_TILE) + 2));  // (also the chained pass's descriptors)
        TRY(dev_realloc(c, c->d_enc_rep, (size_t)n_chunks));
        TRY(dev_realloc(c, c->d_enc_mid, (size_t)((n_chunks + ENC_THREADS - 1) / ENC_THREADS) * ENC_THREADS));
        TRY(dev_realloc(c, c->d_enc_midn, (size_t)((n_chunks + ENC_THREADS - 1) / ENC_THREADS)));
        c->cap_enc_chunks = n_chunks;
    }
    int32_t *d_out = resident ? ids_out : c->d_enc_out;
    unsigned long long *d_ooff = resident ? (unsigned long long *)out_offsets : c->d_enc_off;
    // the chunk cache (k_encode.hip): a slot per four chunks, 2^12 .. 2^22 slots of 32 bytes (the distinct
    // chunks of a text are few; what does not fit is encoded on its own); chunk indices are 32-bit there
    const bool cache = c->enc_cache && n_chunks < 0xFFFFFFFFull;
    uint64_t tslots = 1ull << 12;
    while (tslots < n_chunks / 4 && tslots < (1ull << 22)) tslots <<= 1;
    if (cache) {
        if (tslots > c->cap_enc_tab) {
            TRY(dev_realloc(c, c->d_enc_tab, (size_t)tslots));
            c->cap_enc_tab = tslots;
        }
        // (key 0 = empty; rep all ones: the atomicMin of the hashed chunks starts from there)
        HIPCHK(c, hipMemsetAsync(c->d_enc_tab, 0, tslots * sizeof(EncEntry), c->stream));
        hipLaunchKernelGGL(k_enc_tab_init, dim3((unsigned)((tslots + 255) / 256)), dim3(256), 0, c->stream, c->d_enc_tab,
                           tslots);
        LAUNCHCHK(c, "k_enc_tab_init");
    }
    unsigned long long *d_nlong = c->d_scratch, *d_total = c->d_scratch + 1, *d_nhuge = c->d_scratch + 4;
    uint32_t *d_min = (uint32_t *)(c->d_scratch + 2);
    HIPCHK(c, hipMemsetAsync(c->d_scratch, 0, 2 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(d_nhuge, 0, sizeof(unsigned long long), c->stream));
    const uint32_t mask = (uint32_t)(hs - 1);
    // 4. one chunk per lane -- with the cache, one DISTINCT chunk per lane
    TRY(prof_begin(c, BPE_PROF_ENCODE, n));
    const unsigned gch = (unsigned)((n_chunks + ENC_THREADS - 1) / ENC_THREADS);
    if (cache) {
        const unsigned long long keep = c->enc_hash_bits ? ((1ull << c->enc_hash_bits) - 1ull) << 20 : ~0ull;
        const uint32_t tmask = (uint32_t)(tslots - 1);
        hipLaunchKernelGGL(k_enc_pass1, dim3(gch), dim3(ENC_THREADS), 0, c->stream, c->d_bytes, d_offs, n_chunks, n,
                           c->d_enc_tab, tmask, c->d_enc_rep, keep, c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp,
                           c->d_enc_len, c->d_enc_long, d_nlong, c->d_enc_mid, c->d_enc_midn);
        hipLaunchKernelGGL(k_enc_pass2, dim3(gch), dim3(ENC_THREADS), 0, c->stream, c->d_bytes, d_offs, n_chunks, n,
                           c->d_enc_tab, c->d_enc_rep, c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp,
                           c->d_enc_len, c->d_enc_mid, c->d_enc_midn);
    } else if (bpe_encode_uses_16bit(merge_ids, M))  // 16-bit token / rank columns when every id and every rank fits (rank 0xFFFF = "none")
        hipLaunchKernelGGL(k_encode_short<uint16_t>, dim3(gch),
                           dim3(ENC_THREADS), 0, c->stream, c->d_bytes, d_offs, n_chunks, n,
                           c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp, c->d_enc_len,
                           c->d_enc_long, d_nlong);
    else
        hipLaunchKernelGGL(k_encode_short<uint32_t>, dim3(gch),
                           dim3(ENC_THREADS), 0, c->stream, c->d_bytes, d_offs, n_chunks, n,
                           c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp, c->d_enc_len,
                           c->d_enc_long, d_nlong);
    LAUNCHCHK(c, "k_encode_short");
    // 5. chunks of more than ENC_LMAX bytes: one wave per chunk with the chunk in LDS (k_enc_long), straight off the list
    // pass 1 made -- its length stays on the device; what is longer than ENC_LONG_TOP goes on to d_enc_huge
    if (c->enc_long) {
        const unsigned glong = (unsigned)std::min<uint64_t>(n / (ENC_LMAX + 1) + 1, (uint64_t)c->num_cus * 16);
        hipLaunchKernelGGL((k_enc_long<ENC_LONG_MID, 64>), dim3(glong), dim3(64), 0, c->stream, c->d_bytes, d_offs, n_chunks, n,
                           c->d_enc_long, d_nlong, c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp, c->d_enc_len,
                           c->d_enc_huge, d_nhuge, (uint32_t)ENC_LMAX);
        hipLaunchKernelGGL((k_enc_long<ENC_LONG_MAX, 256>), dim3(std::min(glong, (unsigned)c->num_cus * 2)), dim3(256), 0, c->stream,
                           c->d_bytes, d_offs, n_chunks, n, c->d_enc_long, d_nlong, c->d_ht_keys, c->d_ht_vals, mask, d_mids,
                           c->d_enc_tmp, c->d_enc_len, c->d_enc_huge, d_nhuge, (uint32_t)ENC_LONG_MID);
        // (a whole CU's LDS per chunk: one workgroup per CU)
        hipLaunchKernelGGL((k_enc_long<ENC_LONG_TOP, 1024>), dim3(std::min(glong, (unsigned)c->num_cus)), dim3(1024), 0, c->stream,
                           c->d_bytes, d_offs, n_chunks, n, c->d_enc_long, d_nlong, c->d_ht_keys, c->d_ht_vals, mask, d_mids,
                           c->d_enc_tmp, c->d_enc_len, c->d_enc_huge, d_nhuge, (uint32_t)ENC_LONG_MAX);
        LAUNCHCHK(c, "k_enc_long");
    }
    TRY(prof_end(c));
    unsigned long long n_long = 0;
    HIPCHK(c, hipMemcpyAsync(&n_long, c->enc_long ? d_nhuge : d_nlong, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // ... and the chunks beyond that (a BasicTokenizer text is ONE chunk): stream-wide rounds (lowest rank present ->
    // merge everywhere)
    if (n_long) {
        const unsigned long long *d_list = c->enc_long ? c->d_enc_huge : c->d_enc_long;
        std::vector<unsigned long long> ids_l(n_long);
        HIPCHK(c, hipMemcpy(ids_l.data(), d_list, n_long * 8, hipMemcpyDeviceToHost));
        std::sort(ids_l.begin(), ids_l.end());
        std::vector<unsigned long long> src(n_long), dst(n_long + 1);
        unsigned long long tot = 0;
        std::vector<unsigned long long> range(2 * n_long);
        if (resident) {  // (the caller's offsets live on the device: the chunks' byte ranges in one kernel and one copy)
            DevTmp t_list, t_range;
            HIPCHK(c, t_list.alloc(n_long * 8));
            HIPCHK(c, t_range.alloc(2 * n_long * 8));
            HIPCHK(c, hipMemcpyAsync(t_list.as<unsigned long long>(), ids_l.data(), n_long * 8, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_long_ranges, dim3((unsigned)((n_long + 255) / 256)), dim3(256), 0, c->stream, d_offs, n_chunks, n,
                               t_list.as<unsigned long long>(), (uint64_t)n_long, t_range.as<unsigned long long>());
            LAUNCHCHK(c, "k_long_ranges");
            HIPCHK(c, hipMemcpyAsync(range.data(), t_range.as<unsigned long long>(), 2 * n_long * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        } else {
            for (uint64_t k = 0; k < n_long; k++) {
                const uint64_t ch = ids_l[k];
                range[2 * k] = chunk_offsets[ch];
                range[2 * k + 1] = ch + 1 < n_chunks ? chunk_offsets[ch + 1] : n;
            }
        }
        for (uint64_t k = 0; k < n_long; k++) {
            const uint64_t s0 = range[2 * k], e0 = range[2 * k + 1];
            src[k] = s0;
            dst[k] = tot;
            tot += e0 - s0;
        }
        dst[n_long] = tot;
        TRY(ensure_table(c, 256));
        TRY(ensure_ids(c, tot));
        DevTmp t_src, t_dst, t_cid, t_starts;
        HIPCHK(c, t_src.alloc(n_long * 8));
        HIPCHK(c, t_dst.alloc((n_long + 1) * 8));
        HIPCHK(c, t_cid.alloc(n_long * 8));
        HIPCHK(c, t_starts.alloc((n_long + 1) * 8));
        unsigned long long *d_src = t_src.as<unsigned long long>(), *d_dst = t_dst.as<unsigned long long>(),
                           *d_cid = t_cid.as<unsigned long long>(), *d_starts = t_starts.as<unsigned long long>();
        HIPCHK(c, hipMemcpyAsync(d_src, src.data(), n_long * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_dst, dst.data(), (n_long + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_cid, ids_l.data(), n_long * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_long_gather, dim3((unsigned)n_long), dim3(256), 0, c->stream, c->d_bytes,
                           d_src, d_dst, (uint64_t)n_long, c->d_ids[0]);
        LAUNCHCHK(c, "k_long_gather");
        hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1), 0, c->stream, c->d_st, tot);
        c->apply_target = 0;
        c->par = 0;
        c->n = tot;
        int rc_long = BPE_OK;
        for (;;) {
            HIPCHK(c, hipMemsetAsync(d_min, 0xFF, 4, c->stream));
            if (c->n >= 2) {
                hipLaunchKernelGGL(k_min_rank, dim3(grid_for(c->n, 256, c->num_cus * 8)), dim3(256), 0,
                                   c->stream, c->d_ids[c->par], c->d_st, c->par, c->d_ht_keys,
                                   c->d_ht_vals, mask, d_min);
                LAUNCHCHK(c, "k_min_rank");
            }
            uint32_t r = 0xFFFFFFFFu;
            HIPCHK(c, hipMemcpyAsync(&r, d_min, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (r == 0xFFFFFFFFu) break;
            const int32_t newid = merge_ids ? merge_ids[r] : 256 + (int32_t)r;
            hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, c->stream, c->d_st, merges[2 * r],
                               merges[2 * r + 1]);
            if ((rc_long = launch_merge(c, (uint32_t)newid, 0, nullptr, false)) != BPE_OK) break;
            DevState stt;
            if ((rc_long = read_state(c, &stt)) != BPE_OK) break;
            c->n = stt.n[c->par];
        }
        if (rc_long == BPE_OK) {
            HIPCHK(c, hipMemsetAsync(c->d_scratch + 3, 0, 8, c->stream));
            hipLaunchKernelGGL(k_collect_starts, dim3(grid_for(c->n, 256, c->num_cus * 8)), dim3(256), 0,
                               c->stream, c->d_ids[c->par], c->n, d_starts, (unsigned long long)n_long,
                               c->d_scratch + 3);
            std::vector<unsigned long long> starts(n_long + 1);
            HIPCHK(c, hipMemcpyAsync(starts.data(), d_starts, n_long * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::sort(starts.begin(), starts.begin() + n_long);
            starts[n_long] = c->n;
            HIPCHK(c, hipMemcpyAsync(d_starts, starts.data(), (n_long + 1) * 8, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_long_scatter, dim3((unsigned)n_long), dim3(256), 0, c->stream,
                               c->d_ids[c->par], d_starts, d_cid, d_src, (uint64_t)n_long, c->d_enc_tmp,
                               c->d_enc_len);
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (rc_long != BPE_OK) return rc_long;
    }
    // 6. output offsets = exclusive scan of the per-chunk lengths (cached chunks take their owner's first)
    TRY(prof_begin(c, BPE_PROF_ENCODE, 0));
    if (cache) {
        // (token counts, the scan over them and the placement in one chained pass; option enc_chain = 0: the
        // three-launch form -- counts, block sums, scan inside the tile + placement)
        if (c->enc_chain) {
            const uint64_t nbp = (n_chunks + ENC_PLACE_TILE - 1) / ENC_PLACE_TILE;
            HIPCHK(c, hipMemsetAsync(c->d_enc_bsum, 0, (nbp + 1) * sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(d_min, 0, 4, c->stream));  // (the tile ticket; the long-chunk rounds are over)
            hipLaunchKernelGGL(k_enc_place_chained, dim3((unsigned)nbp), dim3(256), 0, c->stream, c->d_enc_tmp, d_offs,
                               c->d_enc_tab, c->d_enc_rep, c->d_enc_len, c->d_enc_bsum, d_min, d_ooff, n_chunks,
                               d_out, d_total);
        } else {
            hipLaunchKernelGGL(k_enc_lens, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_tab, c->d_enc_rep, n_chunks,
                               c->d_enc_len, c->d_enc_bsum);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_enc_bsum, nb, d_total);
            hipLaunchKernelGGL(k_enc_place, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_tmp, d_offs,
                               c->d_enc_tab, c->d_enc_rep, c->d_enc_len, c->d_enc_bsum, d_ooff, n_chunks, d_out);
        }
    } else {
        hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_len, n_chunks,
                           c->d_enc_bsum);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_enc_bsum, nb, d_total);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_len, n_chunks,
                           c->d_enc_bsum, d_ooff);
        hipLaunchKernelGGL(k_encode_place, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_enc_tmp, d_offs, c->d_enc_len, d_ooff, n_chunks, d_out);
    }
    LAUNCHCHK(c, "k_encode_place");
    TRY(prof_end(c));
    unsigned long long total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (resident) {
        hipLaunchKernelGGL(k_store_u64, dim3(1), dim3(1), 0, c->stream, out_offsets + n_chunks, total);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        if (ids_out && total) TRY(download_d2h(c, ids_out, c->d_enc_out, total * sizeof(int32_t)));
        if (out_offsets) {
            TRY(download_d2h(c, out_offsets, c->d_enc_off, n_chunks * 8));
            out_offsets[n_chunks] = total;
        }
    }
    if (n_out) *n_out = total;
    TRY(prof_drain(c));
    return BPE_OK;
}
}  // namespace

extern "C" int bpe_encode_batch(bpe_ctx *c, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                                const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                                uint64_t n_chunks, int32_t *ids_out, uint64_t *out_offsets,
                                uint64_t *n_out) {
    return encode_batch_impl(c, merges, merge_ids, M, bytes, n, chunk_offsets, n_chunks, ids_out, out_offsets, n_out, false);
}

extern "C" int bpe_encode_batch_resident(bpe_ctx *c, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                                         const uint8_t *d_bytes, uint64_t n, const uint64_t *d_chunk_offsets,
                                         uint64_t n_chunks, int32_t *d_ids_out, uint64_t *d_out_offsets,
                                         uint64_t *n_out) {
    return encode_batch_impl(c, merges, merge_ids, M, d_bytes, n, d_chunk_offsets, n_chunks, d_ids_out, d_out_offsets, n_out,
                             true);
}
