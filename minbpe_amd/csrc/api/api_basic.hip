// api_basic.hip -- lifetime, options, input, the single-step functions.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// ============================================================================
extern "C" {

const char *bpe_version(void) { return "minbpe_amd libbpe_hip 0.1 (gfx950)"; }

const char *bpe_last_error(bpe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int bpe_create(int device_id, bpe_ctx **out) {
    if (!out) return fail(nullptr, BPE_E_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, BPE_E_HIP, "no HIP device available: %s",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, BPE_E_ARG, "device %d out of range (%d devices)", device_id, ndev);
    bpe_ctx *c = new bpe_ctx();
    c->device = device_id;
    auto bail = [&](const char *what, hipError_t er) {
        int rc = fail(nullptr, BPE_E_HIP, "%s failed: %s", what, hipGetErrorString(er));
        delete c;
        return rc;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return bail("hipGetDeviceProperties", e);
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the lean iterations' kernels hold up to LEAN_LDS_BYTES of static LDS per workgroup (k_lean.hip): that is a
    // gfx950 figure (160 KB per CU); on a part that offers less per workgroup the general path runs instead
    if (prop.sharedMemPerBlock < (size_t)LEAN_LDS_BYTES) c->lean = 0;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    c->own_stream = true;
    for (const void *fn : {(const void *)k_pair_count_lds, (const void *)k_pair_count_bytes, (const void *)k_pair_count_h32})
        if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES)) != hipSuccess)
            return bail("hipFuncSetAttribute(dynamic LDS)", e);
    if ((e = hipFuncSetAttribute((const void *)k_load_count, hipFuncAttributeMaxDynamicSharedMemorySize, LC_LDS_BYTES)) != hipSuccess)
        return bail("hipFuncSetAttribute(dynamic LDS)", e);
    if ((e = hipFuncSetAttribute((const void *)bpe::bpe_g4::k_index_build, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 bpe::bpe_g4::IDX_H * 4)) != hipSuccess)
        return bail("hipFuncSetAttribute(dynamic LDS)", e);
    if ((e = hipMalloc((void **)&c->d_st, sizeof(DevState))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&c->d_scratch, 8 * sizeof(unsigned long long))) != hipSuccess)
        return bail("hipMalloc", e);
    *out = c;
    return BPE_OK;
}

void bpe_destroy(bpe_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)bpe_comm_destroy(c);
    for (ProfEv &ev : c->prof_open) {
        (void)hipEventDestroy(ev.e0);
        (void)hipEventDestroy(ev.e1);
    }
    for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
    void *ptrs[] = {c->d_bytes, c->d_offsets, c->d_ids[0], c->d_ids[1], c->d_mat,  c->d_first,
                    c->d_rowmax, c->d_st,     c->d_tsum,   c->d_tile_off, c->d_tile_sin, c->d_sup, c->d_scratch,
                    c->d_delta,  c->d_dirty_list, c->d_dirty_n, c->d_desc, c->d_gdesc, c->d_enc_tmp, c->d_enc_len, c->d_enc_out,
                    c->d_enc_off, c->d_enc_bsum, c->d_enc_long, c->d_ht_keys, c->d_ht_vals, c->d_merge_ids,
                    c->d_dp_folded, c->d_dp_table, c->d_dp_key, c->d_meta[0], c->d_meta[1], c->d_slot_lens,
                    c->d_slot_off, c->d_slot_bsum, c->d_ids2, c->d_hdr[0], c->d_hdr[1],
                    c->d_dec_blob, c->d_dec_out, c->d_dec_voff, c->d_dec_off, c->d_dec_bsum, c->d_dec_ids,
                    c->d_dec_len, c->d_wexp, c->d_round_lb, c->d_dp_ckey, c->d_dp_cfold, c->d_hdr2[0], c->d_hdr2[1], c->d_stage, c->d_idx, c->d_idx_tmp, c->d_idx_dirty, c->d_removed, c->d_smask, c->d_cand, c->d_dbits, c->d_lean_res, c->d_lean_sum, c->d_enc_tab, c->d_enc_rep, c->d_enc_mid, c->d_enc_midn, c->d_chain_req, c->d_pool, c->d_pool_gather, c->d_enc_huge};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_srec) (void)hipHostFree(c->h_srec);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (hipEvent_t ev : c->ev_stage)
        if (ev) (void)hipEventDestroy(ev);
    for (void *p : {(void *)c->d_step_pub, (void *)c->d_step_bar, (void *)c->d_step_stamps, (void *)c->d_forced})
        if (p) (void)hipFree(p);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bpe_set_stream(bpe_ctx *c, void *hip_stream) {
    if (!c) return BPE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) HIPCHK(c, hipStreamDestroy(c->stream));
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return BPE_OK;
}

int bpe_set_option(bpe_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return BPE_E_ARG;
    if (!strcmp(name, "mode")) {
        if (value != 0 && value != 1) return fail(c, BPE_E_ARG, "mode must be 0 or 1");
        c->mode = (int)value;
    } else if (!strcmp(name, "profile")) {
        if (value < 0 || value > 2) return fail(c, BPE_E_ARG, "profile must be 0, 1 or 2");
        c->profile = (int)value;
    } else if (!strcmp(name, "k1")) {
        c->k1 = (int)value;
    } else if (!strcmp(name, "scan_sup")) {
        if (value < 0 || value > (1 << 30)) return fail(c, BPE_E_ARG, "scan_sup: 0 .. 2^30 tiles");
        c->scan_sup_min = (int)value;
    } else if (!strcmp(name, "merge")) {
        if (value != 0 && value != 1) return fail(c, BPE_E_ARG, "merge must be 0 or 1");
        c->merge_impl = (int)value;
    } else if (!strcmp(name, "lb_tune")) {
        c->lb_tune = (uint32_t)value;
    } else if (!strcmp(name, "fused_rows")) {
        c->fused_rows = value != 0;
    } else if (!strcmp(name, "slots")) {
        if (value < 0 || value > 2) return fail(c, BPE_E_ARG, "slots must be 0, 1 or 2");
        c->use_slots = (int)value;
    } else if (!strcmp(name, "sparse")) {
        if (value < 0 || value > 2) return fail(c, BPE_E_ARG, "sparse must be 0 (never), 1 (auto) or 2 (always)");
        c->use_sparse = (int)value;
    } else if (!strcmp(name, "rep_max")) {
        if (value < 0 || value > 8) return fail(c, BPE_E_ARG, "rep_max must be 0..8");
        c->rep_max = (int)value;
    } else if (!strcmp(name, "rep_min")) {
        if (value < 0 || value > 8) return fail(c, BPE_E_ARG, "rep_min: 0..8");
        c->rep_min = value;
    } else if (!strcmp(name, "lds_delta")) {
        c->lds_delta = value != 0;
    } else if (!strcmp(name, "exp_no_delta")) {
        c->exp_no_delta = value != 0;
    } else if (!strcmp(name, "tie_window")) {
        c->tie_window = value != 0;
    } else if (!strcmp(name, "tie_index")) {
        c->tie_index = value != 0;
    } else if (!strcmp(name, "sparse_ratio")) {
        if (value < 1 || value > 64) return fail(c, BPE_E_ARG, "sparse_ratio must be 1..64");
        c->sparse_ratio = (int)value;
    } else if (!strcmp(name, "lean")) {
        if (value < 0 || value > 2) return fail(c, BPE_E_ARG, "lean must be 0 (never), 1 (auto) or 2 (always)");
        c->lean = (int)value;
    } else if (!strcmp(name, "lean_backoff")) {
        c->lean_backoff = value != 0;
    } else if (!strcmp(name, "lean_count")) {
        if (value < 0) return fail(c, BPE_E_ARG, "lean_count must be >= 0");
        c->lean_count = value;
    } else if (!strcmp(name, "lean_grid")) {
        if (value < 1 || value > 65535) return fail(c, BPE_E_ARG, "lean_grid must be 1..65535");
        c->lean_grid = (int)value;
    } else if (!strcmp(name, "enc_chain")) {
        c->enc_chain = value != 0;
    } else if (!strcmp(name, "enc_cache")) {
        c->enc_cache = value != 0;
    } else if (!strcmp(name, "enc_hash_bits")) {
        if (value < 0 || value > 40) return fail(c, BPE_E_ARG, "enc_hash_bits must be 0..40");
        c->enc_hash_bits = (int)value;
    } else if (!strcmp(name, "prof_stride")) {
        if (value < 1 || value > 1024) return fail(c, BPE_E_ARG, "prof_stride must be 1..1024");
        c->prof_stride = (int)value;
    } else if (!strcmp(name, "lean_select")) {
        c->lean_select = value != 0;
    } else if (!strcmp(name, "aa_sparse")) {
        c->aa_sparse = value != 0;
    } else if (!strcmp(name, "dp_force_comm")) {
        c->dp_force_comm = value != 0;
    } else if (!strcmp(name, "dp_kcap")) {
        if (value < 1 || value > DP_KCAP_MAX) return fail(c, BPE_E_ARG, "dp_kcap must be 1..%d", DP_KCAP_MAX);
        c->dp_kcap = (int)value;
    } else if (!strcmp(name, "fuse_load")) {
        c->fuse_load = value != 0;
    } else if (!strcmp(name, "chain_dense")) {
        c->chain_dense = value != 0;
    } else if (!strcmp(name, "chain_prefetch")) {
        c->chain_prefetch = value != 0;
    } else if (!strcmp(name, "count_is_removed")) {
        c->count_is_removed = value != 0;
    } else if (!strcmp(name, "enc_long")) {
        c->enc_long = value != 0;
    } else if (!strcmp(name, "small_slots")) {
        if (value < 0 || value > 2) return fail(c, BPE_E_ARG, "small_slots: 0, 1 or 2");
        c->small_slots = (int)value;
    } else if (!strcmp(name, "chain_kcap")) {
        if (value < 1 || value > CH_KSWEEP) return fail(c, BPE_E_ARG, "chain_kcap must be 1..%d", CH_KSWEEP);
        c->chain_kcap = (int)value;
    } else if (!strcmp(name, "enc_replay")) {
        if (value < 0 || value > 1) return fail(c, BPE_E_ARG, "enc_replay must be 0 or 1");
        c->enc_replay = (int)value;
    } else if (!strcmp(name, "pinned_upload")) {
        if (value < 0 || value > 1) return fail(c, BPE_E_ARG, "pinned_upload must be 0 or 1");
        c->pinned_upload = (int)value;
    } else if (!strcmp(name, "fuse_step")) {
        if (value < 0 || value > 1) return fail(c, BPE_E_ARG, "fuse_step must be 0 or 1");
        c->fuse_step = (int)value;
    } else if (!strcmp(name, "pool_hint")) {
        if (value < 0 || value > 128) return fail(c, BPE_E_ARG, "pool_hint must be 0..128");
        c->pool_hint = (int)value;
    } else if (!strcmp(name, "chain_scan")) {
        if (value < 1 || value > 255) return fail(c, BPE_E_ARG, "chain_scan must be 1..255");
        c->chain_scan = value;
    } else if (!strcmp(name, "chain")) {
        c->chain = value != 0;
    } else if (!strcmp(name, "lean_chain")) {
        c->lean_chain = value != 0;
    } else if (!strcmp(name, "lean_sum")) {
        c->lean_sum = value != 0;
    } else if (!strcmp(name, "lean_scan")) {
        if (value < 1 || value > 1024) return fail(c, BPE_E_ARG, "lean_scan must be 1..1024");
        c->lean_scan = (int)value;
    } else if (!strcmp(name, "repack_acc")) {
        if (value < 0 || value > 10000) return fail(c, BPE_E_ARG, "repack_acc must be 0..10000");
        c->repack_acc = (int)value;
    } else if (!strcmp(name, "depth")) {
        if (value < 0 || value > 64) return fail(c, BPE_E_ARG, "depth must be 0..64");
        c->depth = (int)value;
    } else {
        return fail(c, BPE_E_ARG, "unknown option '%s'", name);
    }
    return BPE_OK;
}

// chunk start offsets as the header states them: ascending (equal neighbours = an empty chunk),
// none beyond n.  Every entry point that takes offsets checks them: the kernels compute chunk
// lengths as differences and would index out of bounds on a descending pair.
// chunk offsets must ascend and stay <= n.  A GPT-4-split GB has 170 M of them (1.4 GB): one thread took ~40 ms over
// them before the first byte was uploaded; up to 16 threads take a segment each (the first bad index wins, so the message
// is the one the serial scan gave).
static int check_offsets(bpe_ctx *c, const uint64_t *chunk_offsets, uint64_t n_chunks, uint64_t n) {
    if (!chunk_offsets) return BPE_OK;
    auto scan = [&](uint64_t lo, uint64_t hi) -> uint64_t {  // first bad index in [lo, hi), or ~0
        uint64_t prev = lo ? chunk_offsets[lo - 1] : 0;
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t o = chunk_offsets[i];
            if (o < prev || o > n) return i;
            prev = o;
        }
        return ~0ull;
    };
    uint64_t bad = ~0ull;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t T = n_chunks < (1ull << 22) ? 1 : std::max<uint64_t>(1, std::min<uint64_t>(16, hw / 2));
    if (T == 1) {
        bad = scan(0, n_chunks);
    } else {
        std::vector<uint64_t> first(T, ~0ull);
        std::vector<std::thread> pool;
        const uint64_t per = (n_chunks + T - 1) / T;
        for (uint64_t w = 0; w < T; w++)
            pool.emplace_back([&, w] { first[w] = scan(std::min(n_chunks, w * per), std::min(n_chunks, (w + 1) * per)); });
        for (std::thread &t : pool) t.join();
        for (uint64_t w = 0; w < T; w++) bad = std::min(bad, first[w]);
    }
    if (bad != ~0ull)
        return fail(c, BPE_E_ARG, "chunk_offsets[%llu] = %llu: offsets must ascend and stay <= n = %llu",
                    (unsigned long long)bad, (unsigned long long)chunk_offsets[bad], (unsigned long long)n);
    return BPE_OK;
}

static int load_bytes_impl(bpe_ctx *c, const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                           uint64_t n_chunks, const uint8_t *wexp) {
    if (!c || (!bytes && n)) return fail(c, BPE_E_ARG, "bytes is NULL");
    TRY(check_offsets(c, chunk_offsets, n_chunks, n));
    if (n >= (1ull << 32)) return fail(c, BPE_E_LIMIT, "stream of %llu bytes exceeds 2^32-1 per GPU",
                                       (unsigned long long)n);
    if (wexp && !chunk_offsets) return fail(c, BPE_E_ARG, "weights need chunk offsets");
    if (wexp) {  // (checked before anything is queued: no copy may outlive a failed call)
        // pair counts are 32-bit: the text the weighted chunks stand for must stay below 2^32 bytes
        unsigned __int128 stands_for = 0;
        for (uint64_t i = 0; i < n_chunks; i++) {
            if (wexp[i] > 31) return fail(c, BPE_E_ARG, "weight exponent %u of chunk %llu exceeds 31", wexp[i],
                                          (unsigned long long)i);
            const uint64_t b = chunk_offsets[i], e = i + 1 < n_chunks ? chunk_offsets[i + 1] : n;
            if (e > b && e <= n) stands_for += (unsigned __int128)(e - b) << wexp[i];
        }
        if (stands_for >= ((unsigned __int128)1 << 32))
            return fail(c, BPE_E_LIMIT, "the weighted chunks stand for >= 2^32 bytes of text (32-bit pair counts)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (n + 16 > c->cap_bytes) {
        TRY(dev_realloc(c, c->d_bytes, (size_t)n + 16));
        c->cap_bytes = n + 16;
    }
    if (n) TRY(upload_h2d(c, c->d_bytes, bytes, n));
    static const uint64_t zero = 0;
    if (!chunk_offsets) {
        chunk_offsets = &zero;
        n_chunks = 1;
    }
    if (n_chunks > c->cap_offsets) {
        TRY(dev_realloc(c, c->d_offsets, (size_t)n_chunks));
        c->cap_offsets = n_chunks;
    }
    if (n_chunks) TRY(upload_h2d(c, c->d_offsets, chunk_offsets, n_chunks * sizeof(uint64_t)));
    c->weighted = false;
    if (wexp && n_chunks) {
        if (n_chunks > c->cap_wexp) {
            TRY(dev_realloc(c, c->d_wexp, (size_t)n_chunks));
            c->cap_wexp = n_chunks;
        }
        TRY(upload_h2d(c, c->d_wexp, wexp, n_chunks));
        c->weighted = true;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // caller may free its buffers on return
    c->nbytes = n;
    c->n_chunks = n_chunks;
    c->have_bytes = true;
    TRY(ensure_table(c, 256));
    TRY(start_from_bytes(c));
    return BPE_OK;
}

int bpe_load_bytes(bpe_ctx *c, const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                   uint64_t n_chunks) {
    return load_bytes_impl(c, bytes, n, chunk_offsets, n_chunks, nullptr);
}

int bpe_load_bytes_weighted(bpe_ctx *c, const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                            uint64_t n_chunks, const uint8_t *weight_exp) {
    if (!weight_exp) return fail(c, BPE_E_ARG, "weight_exp is NULL");
    return load_bytes_impl(c, bytes, n, chunk_offsets, n_chunks, weight_exp);
}

int bpe_load_ids(bpe_ctx *c, const int32_t *ids, uint64_t n, const uint64_t *chunk_offsets,
                 uint64_t n_chunks) {
    if (!c || (!ids && n)) return fail(c, BPE_E_ARG, "ids is NULL");
    if (n >= (1ull << 32)) return fail(c, BPE_E_LIMIT, "stream too long");
    TRY(check_offsets(c, chunk_offsets, n_chunks, n));
    HIPCHK(c, hipSetDevice(c->device));
    int32_t mx = 255;
    for (uint64_t i = 0; i < n; i++) {
        if (ids[i] < 0) return fail(c, BPE_E_ARG, "negative token id at %llu", (unsigned long long)i);
        mx = std::max(mx, ids[i]);
    }
    TRY(ensure_table(c, (uint32_t)mx + 1));
    TRY(ensure_ids(c, n));
    // stage through buffer 1, then mask into buffer 0
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_ids[1], ids, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_load_ids, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                           (const int32_t *)c->d_ids[1], c->d_ids[0], n);
        LAUNCHCHK(c, "k_load_ids");
    }
    static const uint64_t zero = 0;
    if (!chunk_offsets) {
        chunk_offsets = &zero;
        n_chunks = 1;
    }
    if (n_chunks > c->cap_offsets) {
        TRY(dev_realloc(c, c->d_offsets, (size_t)n_chunks));
        c->cap_offsets = n_chunks;
    }
    if (n && n_chunks) {
        HIPCHK(c, hipMemcpyAsync(c->d_offsets, chunk_offsets, n_chunks * sizeof(uint64_t),
                                 hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(n_chunks, 256, c->num_cus * 8)), dim3(256), 0,
                           c->stream, c->d_ids[0], c->d_offsets, n_chunks, n);
        LAUNCHCHK(c, "k_mark_starts");
    }
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1), 0, c->stream, c->d_st, (unsigned long long)n);
    LAUNCHCHK(c, "k_init_state");
    c->apply_target = 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_bytes = false;
    c->weighted = false;
    c->stream_is_bytes = false;
    c->par = 0;
    c->n = n;
    c->vcur = (uint32_t)mx + 1;
    c->have_ids = true;
    c->stats_valid = false;
    return BPE_OK;
}

int bpe_get_stats(bpe_ctx *c, uint64_t *n_pairs_out) {
    if (!c) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_first) TRY(dev_realloc(c, c->d_first, (size_t)c->vcap * c->vcap));
    TRY(clear_table(c));
    HIPCHK(c, hipMemsetAsync(c->d_first, 0xFF, (size_t)c->vcur * c->vcap * sizeof(uint32_t), c->stream));
    TRY(launch_pair_count(c, true));
    HIPCHK(c, hipMemsetAsync(c->d_scratch, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_count_nonzero, dim3(c->vcur), dim3(256), 0, c->stream, c->d_mat, c->vcap,
                       c->vcur, c->d_scratch);
    LAUNCHCHK(c, "k_count_nonzero");
    unsigned long long np = 0;
    HIPCHK(c, hipMemcpyAsync(&np, c->d_scratch, sizeof np, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stats_valid = true;
    if (n_pairs_out) *n_pairs_out = np;
    return BPE_OK;
}

int bpe_read_stats(bpe_ctx *c, int32_t *a, int32_t *b, uint64_t *cnt, uint64_t *first_pos,
                   uint64_t cap, uint64_t *n_out) {
    if (!c) return BPE_E_ARG;
    if (!c->stats_valid) return fail(c, BPE_E_STATE, "bpe_get_stats has not been run on the current ids");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp ta, tb, tc, tf;
    const size_t capn = cap ? cap : 1;
    HIPCHK(c, ta.alloc(capn * 4));
    HIPCHK(c, tb.alloc(capn * 4));
    HIPCHK(c, tc.alloc(capn * 8));
    HIPCHK(c, tf.alloc(capn * 8));
    int32_t *da = ta.as<int32_t>(), *db = tb.as<int32_t>();
    unsigned long long *dc = tc.as<unsigned long long>(), *df = tf.as<unsigned long long>();
    HIPCHK(c, hipMemsetAsync(c->d_scratch, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_dump_stats, dim3(c->vcur), dim3(256), 0, c->stream, c->d_mat, c->d_first,
                       c->vcap, c->vcur, da, db, dc, df, (unsigned long long)cap, c->d_scratch);
    LAUNCHCHK(c, "k_dump_stats");
    unsigned long long np = 0;
    HIPCHK(c, hipMemcpyAsync(&np, c->d_scratch, sizeof np, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = BPE_OK;
    if (np > cap) {
        rc = fail(c, BPE_E_CAP, "%llu pairs but cap is %llu", np, (unsigned long long)cap);
    } else if (np) {
        HIPCHK(c, hipMemcpy(a, da, np * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(b, db, np * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(cnt, dc, np * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(first_pos, df, np * 8, hipMemcpyDeviceToHost));
    }
    if (n_out) *n_out = np;
    return rc;
}

int bpe_argmax(bpe_ctx *c, int32_t *a, int32_t *b, uint64_t *count) {
    if (!c) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, c->stream, c->d_st, 0, 0);  // clears status
    TRY(clear_table(c));
    TRY(launch_pair_count(c, false));
    c->stats_valid = false;
    TRY(launch_select(c, true));
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, stream_ref(c), c->par, c->d_st);
    LAUNCHCHK(c, "k_finalize");
    DevState st;
    TRY(read_state(c, &st));
    if (st.status == ST_EMPTY) return fail(c, BPE_E_EMPTY_STATS, "max() arg is an empty sequence");
    if (st.status != ST_OK) return fail(c, BPE_E_INTERNAL, "device status %u", st.status);
    if (a) *a = st.a;
    if (b) *b = st.b;
    if (count) *count = st.count;
    return BPE_OK;
}

int bpe_merge(bpe_ctx *c, int32_t a, int32_t b, int32_t idx, uint64_t *new_len) {
    if (!c) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    if (a < 0 || b < 0 || idx < 0) return fail(c, BPE_E_ARG, "negative id");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_table(c, (uint32_t)std::max(idx, std::max(a, b)) + 1));
    hipLaunchKernelGGL(k_set_pair, dim3(1), dim3(1), 0, c->stream, c->d_st, a, b);
    LAUNCHCHK(c, "k_set_pair");
    const uint64_t n_before = c->n;
    TRY(launch_merge(c, (uint32_t)idx, 0, nullptr, false));
    DevState st;
    TRY(read_state(c, &st));
    c->n = st.n[c->par];
    if (c->profile) c->prof_bytes[BPE_PROF_MERGE] += 4 * (n_before + c->n);
    c->vcur = std::max<uint32_t>(c->vcur, (uint32_t)idx + 1);
    if (new_len) *new_len = c->n;
    return BPE_OK;
}

int bpe_len(bpe_ctx *c, uint64_t *n) {
    if (!c || !n) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    *n = c->n;
    return BPE_OK;
}

int bpe_read_ids(bpe_ctx *c, int32_t *out, uint64_t cap) {
    if (!c) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    if (cap < c->n) return fail(c, BPE_E_CAP, "need %llu entries", (unsigned long long)c->n);
    if (!c->n) return BPE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // strip flags into the idle ping-pong buffer, then copy out
    int32_t *tmp = (int32_t *)c->d_ids[c->par ^ 1];
    hipLaunchKernelGGL(k_strip_flags, dim3(grid_for(c->n, 256, c->num_cus * 8)), dim3(256), 0,
                       c->stream, c->d_ids[c->par], tmp, c->n);
    LAUNCHCHK(c, "k_strip_flags");
    HIPCHK(c, hipMemcpyAsync(out, tmp, c->n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BPE_OK;
}

int bpe_read_chunk_starts(bpe_ctx *c, uint64_t *out, uint64_t cap, uint64_t *n_out) {
    if (!c) return BPE_E_ARG;
    if (!c->have_ids) return fail(c, BPE_E_STATE, "no ids loaded");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp t_out;
    HIPCHK(c, t_out.alloc((cap ? cap : 1) * 8));
    unsigned long long *d_out = t_out.as<unsigned long long>();
    HIPCHK(c, hipMemsetAsync(c->d_scratch, 0, sizeof(unsigned long long), c->stream));
    if (c->n) {
        hipLaunchKernelGGL(k_collect_starts, dim3(grid_for(c->n, 256, c->num_cus * 8)), dim3(256), 0,
                           c->stream, c->d_ids[c->par], c->n, d_out, (unsigned long long)cap,
                           c->d_scratch);
        LAUNCHCHK(c, "k_collect_starts");
    }
    unsigned long long ns = 0;
    HIPCHK(c, hipMemcpyAsync(&ns, c->d_scratch, sizeof ns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = BPE_OK;
    if (ns > cap) {
        rc = fail(c, BPE_E_CAP, "%llu chunk starts but cap is %llu", ns, (unsigned long long)cap);
    } else if (ns) {
        HIPCHK(c, hipMemcpy(out, d_out, ns * 8, hipMemcpyDeviceToHost));
        std::sort(out, out + ns);
    }
    if (n_out) *n_out = ns;
    return rc;
}

}  // extern "C"
