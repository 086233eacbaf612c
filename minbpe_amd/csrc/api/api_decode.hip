// api_decode.hip -- bpe_decode_*.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// ---------------------------------------------------------------------------
// decode (N4)

extern "C" int bpe_decode_set_vocab(bpe_ctx *c, const uint8_t *vocab_bytes, const uint64_t *vocab_offsets,
                                    int32_t V) {
    if (!c || V < 0 || !vocab_offsets) return fail(c, BPE_E_ARG, "bad arguments");
    if (vocab_offsets[0] != 0) return fail(c, BPE_E_ARG, "vocab_offsets[0] must be 0");
    for (int32_t i = 0; i < V; i++)
        if (vocab_offsets[i + 1] < vocab_offsets[i])
            return fail(c, BPE_E_ARG, "vocab_offsets must not decrease (entry %d)", i);
    const uint64_t nb = vocab_offsets[V];
    if (nb && !vocab_bytes) return fail(c, BPE_E_ARG, "vocab_bytes is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    c->dec_have_vocab = false;
    c->dec_have_result = false;
    if (nb + 16 > c->cap_dec_blob) {
        TRY(dev_realloc(c, c->d_dec_blob, (size_t)nb + 16));
        c->cap_dec_blob = nb + 16;
    }
    if ((uint64_t)V + 1 > c->cap_dec_voff) {
        TRY(dev_realloc(c, c->d_dec_voff, (size_t)V + 1));
        c->cap_dec_voff = (uint64_t)V + 1;
    }
    if (nb) HIPCHK(c, hipMemcpyAsync(c->d_dec_blob, vocab_bytes, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_dec_voff, vocab_offsets, ((size_t)V + 1) * sizeof(uint64_t),
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // caller may free its buffers on return
    c->dec_V = (uint32_t)V;
    c->dec_have_vocab = true;
    return BPE_OK;
}

extern "C" int bpe_decode_batch(bpe_ctx *c, const int32_t *ids, uint64_t n, uint64_t *n_bytes,
                                uint64_t *bad_index) {
    if (!c || (!ids && n)) return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->dec_have_vocab) return fail(c, BPE_E_STATE, "bpe_decode_set_vocab first");
    if (n_bytes) *n_bytes = 0;
    if (bad_index) *bad_index = ~0ull;
    c->dec_have_result = false;
    c->dec_n = n;
    c->dec_total = 0;
    if (n == 0) {
        c->dec_have_result = true;
        return BPE_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (n > c->cap_dec_n) {
        TRY(dev_realloc(c, c->d_dec_ids, (size_t)n));
        TRY(dev_realloc(c, c->d_dec_len, (size_t)n));
        TRY(dev_realloc(c, c->d_dec_off, (size_t)n + 1));
        TRY(dev_realloc(c, c->d_dec_bsum, (size_t)nb + 1));
        c->cap_dec_n = n;
    }
    unsigned long long *d_bad = c->d_scratch, *d_total = c->d_scratch + 1;
    HIPCHK(c, hipMemcpyAsync(c->d_dec_ids, ids, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_bad, 0xFF, sizeof(unsigned long long), c->stream));
    TRY(prof_begin(c, BPE_PROF_DECODE, 4 * n));
    hipLaunchKernelGGL(k_decode_len, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                       c->d_dec_ids, n, c->d_dec_voff, c->dec_V, c->d_dec_len, d_bad);
    LAUNCHCHK(c, "k_decode_len");
    hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_dec_len, n,
                       c->d_dec_bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_dec_bsum, nb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_dec_len, n,
                       c->d_dec_bsum, c->d_dec_off);
    LAUNCHCHK(c, "k_scan_*");
    TRY(prof_end(c));
    unsigned long long hb[2] = {0, 0};  // {first bad position, total bytes}
    HIPCHK(c, hipMemcpyAsync(hb, c->d_scratch, sizeof hb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hb[0] != ~0ull) {
        if (bad_index) *bad_index = hb[0];
        TRY(prof_drain(c));
        return fail(c, BPE_E_ARG, "invalid token id: %d (position %llu)", ids[hb[0]], hb[0]);
    }
    const uint64_t total = hb[1];
    if (total + 16 > c->cap_dec_out) {
        TRY(dev_realloc(c, c->d_dec_out, (size_t)total + 16));
        c->cap_dec_out = total + 16;
    }
    TRY(prof_begin(c, BPE_PROF_DECODE, total));
    hipLaunchKernelGGL(k_decode_copy, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                       c->d_dec_ids, n, c->d_dec_voff, c->dec_V, c->d_dec_blob, c->d_dec_off, c->d_dec_out);
    LAUNCHCHK(c, "k_decode_copy");
    TRY(prof_end(c));
    TRY(prof_drain(c));
    c->dec_total = total;
    c->dec_have_result = true;
    if (n_bytes) *n_bytes = total;
    return BPE_OK;
}

extern "C" int bpe_decode_read(bpe_ctx *c, uint8_t *out, uint64_t cap, const uint64_t *doc_token_offsets,
                               uint64_t k, uint64_t *doc_byte_offsets_out) {
    if (!c) return BPE_E_ARG;
    if (!c->dec_have_result) return fail(c, BPE_E_STATE, "bpe_decode_batch first");
    if (c->dec_total && (!out || cap < c->dec_total))
        return fail(c, BPE_E_CAP, "need %llu bytes", (unsigned long long)c->dec_total);
    if (k && (!doc_token_offsets || !doc_byte_offsets_out)) return fail(c, BPE_E_ARG, "offset arrays are NULL");
    for (uint64_t j = 0; j < k; j++)
        if (doc_token_offsets[j] > c->dec_n)
            return fail(c, BPE_E_ARG, "doc_token_offsets[%llu] is past the last token", (unsigned long long)j);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->dec_total)
        HIPCHK(c, hipMemcpyAsync(out, c->d_dec_out, c->dec_total, hipMemcpyDeviceToHost, c->stream));
    if (k) {
        if (c->dec_n == 0) {  // nothing was decoded: every offset is 0
            for (uint64_t j = 0; j < k; j++) doc_byte_offsets_out[j] = 0;
        } else {
            DevTmp t_idx, t_dst;
            HIPCHK(c, t_idx.alloc(k * 8));
            HIPCHK(c, t_dst.alloc(k * 8));
            HIPCHK(c, hipMemcpyAsync(t_idx.p, doc_token_offsets, k * 8, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_decode_doc_offsets, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream,
                               c->d_dec_off, c->dec_n, (unsigned long long)c->dec_total, t_idx.as<unsigned long long>(), k,
                               t_dst.as<unsigned long long>());
            LAUNCHCHK(c, "k_decode_doc_offsets");
            HIPCHK(c, hipMemcpyAsync(doc_byte_offsets_out, t_dst.p, k * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BPE_OK;
}
