// api_dp.hip -- bpe_dp_*: data-parallel stepping.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// ---------------------------------------------------------------------------
// data-parallel stepping: one ctx per rank, the host runs the two all-reduces

extern "C" int bpe_dp_begin(bpe_ctx *c, int32_t num_merges, int32_t rank, int32_t nranks) {
    if (!c || num_merges < 0 || rank < 0 || nranks < 1 || rank >= nranks || nranks > 1024)
        return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->have_bytes) return fail(c, BPE_E_STATE, "bpe_load_bytes first");
    HIPCHK(c, hipSetDevice(c->device));
    c->dp_rank = rank;
    c->dp_nranks = nranks;
    c->dp_merges = num_merges;
    c->dp_active = true;
    TRY(ensure_table(c, 256u + (uint32_t)num_merges));
    // (a pass that a device status cut short in an earlier call may have left partial sums behind)
    HIPCHK(c, hipMemsetAsync(c->d_delta, 0, ((size_t)c->vcap * 4 * DELTA_REPL + 256 * DELTA_SKEW) * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_rowmax, 0, (size_t)c->vcap * 2 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dbits, 0, DBITS_WORDS * sizeof(uint32_t), c->stream));
    TRY(ensure_rec(c, std::max(num_merges, 1)));
    memset(c->h_rec, 0, sizeof(IterRec) * (size_t)std::max(num_merges, 1));
    if (c->d_dp_folded) (void)hipFree(c->d_dp_folded);
    c->d_dp_folded = nullptr;
    HIPCHK(c, hipMalloc((void **)&c->d_dp_folded, ((size_t)c->vcap * 4 + 64) * sizeof(uint32_t)));
    HIPCHK(c, hipMemsetAsync(c->d_dp_folded, 0, ((size_t)c->vcap * 4 + 64) * sizeof(uint32_t), c->stream));
    if (!c->d_dp_table) HIPCHK(c, hipMalloc((void **)&c->d_dp_table, 2 * 256 * 256 * sizeof(uint32_t)));
    if (!c->d_dp_key) HIPCHK(c, hipMalloc((void **)&c->d_dp_key, 3 * sizeof(long long)));
    HIPCHK(c, hipMemsetAsync(c->d_mat, 0, (size_t)c->vcap * c->vcap * sizeof(uint32_t), c->stream));
    const bool fused_load = load_count_fusable(c);
    TRY(start_from_bytes(c, fused_load));
    if (!fused_load) TRY(launch_pair_count(c, false));
    // the byte-pair block of the table, as 16-bit limbs (the sum over the ranks cannot wrap), is the first all-reduce payload
    hipLaunchKernelGGL(k_dp_table_split, dim3(256), dim3(256), 0, c->stream, c->d_mat, c->vcap, c->d_dp_table);
    LAUNCHCHK(c, "k_dp_table_split");
    c->dp_cur_len = c->n;
    c->dp_enq = c->dp_done = 0;
    c->rep_shift = 5;
    c->idx_live = false;
    c->idx_rebuild = false;
    c->last_count = ~0ull;
    c->n_sparse = c->n_dense = c->n_index_builds = 0;
    c->dp_flip.assign((size_t)std::max(num_merges, 1), 0);
    c->ts = TILE2_MAX;
    if (c->use_slots) TRY(c->use_slots == 2 ? slots2_enter(c) : slots_enter(c));
    return BPE_OK;
}

extern "C" int bpe_dp_buffers(bpe_ctx *c, void **table, uint64_t *table_count, void **delta,
                              uint64_t *delta_count, void **tiekey) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    if (table) *table = c->d_dp_table;
    if (table_count) *table_count = 2 * 256 * 256;  // (low limbs, then high limbs)
    if (delta) *delta = c->d_dp_folded;
    if (delta_count) *delta_count = (uint64_t)c->vcap * 4 + 64;  // four vectors + the format-B adj word (padded)
    if (tiekey) *tiekey = c->d_dp_key;
    return BPE_OK;
}

extern "C" int bpe_dp_table_ready(bpe_ctx *c) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    // the GLOBAL counts from the summed limbs; a pair that occurs 2^32 times or more in the whole job does not fit the
    // 32-bit table -- every rank sees the same sums, so every rank fails here, before any further collective
    HIPCHK(c, hipMemsetAsync(c->d_scratch + 5, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_dp_table_join, dim3(256), dim3(256), 0, c->stream, c->d_dp_table, c->d_mat, c->vcap, c->d_scratch + 5);
    LAUNCHCHK(c, "k_dp_table_join");
    unsigned long long over = 0;
    HIPCHK(c, hipMemcpyAsync(&over, c->d_scratch + 5, sizeof over, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (over)
        return fail(c, BPE_E_LIMIT, "a byte pair occurs %llu times in the %d shards together: counts are 32-bit "
                    "(every rank stops here)", over, c->dp_nranks);
    c->vcur = 256;
    hipLaunchKernelGGL(k_rowmax_all, dim3(256), dim3(256), 0, c->stream, c->d_mat, c->vcap, 256u,
                       c->d_rowmax);
    LAUNCHCHK(c, "k_rowmax_all");
    return BPE_OK;
}

extern "C" int bpe_dp_select(bpe_ctx *c, int32_t iter) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    c->vcur = 256u + (uint32_t)iter;
    const uint64_t den = c->idx_live ? 8 : REPACK_DEN;
    if (c->slotted && c->slot_T > 64 && c->n * den < c->slot_T * (uint64_t)(c->slot2 ? c->ts : TILE) * (den - 1)) {
        if (c->slot2) {
            TRY(slots2_leave(c));
            TRY(slots2_enter(c));
        } else {
            TRY(slots_leave(c));
            TRY(slots_enter(c));
        }
    }
    c->dp_sparse = false;
    if (c->slotted && c->slot2) TRY(plan_pass2(c, &c->dp_sparse));  // (counts are global: every rank decides alike)
    TRY(flush_lean_rows(c, c->vcur));  // (row maxima a chain step's table update left to do: dp_train_loop)
    TRY(launch_select(c, false));
    if (c->slotted && c->slot2)
        hipLaunchKernelGGL(GK(c, k_dp_key<SlotRefH>), dim3(1), dim3(64), 0, c->stream, stream_ref_h(c), c->par, c->d_st,
                           (unsigned long long)c->dp_rank, c->d_dp_key);
    else
        hipLaunchKernelGGL(k_dp_key<SlotRef>, dim3(1), dim3(64), 0, c->stream, stream_ref(c), c->par, c->d_st,
                           (unsigned long long)c->dp_rank, c->d_dp_key);
    LAUNCHCHK(c, "k_dp_key");
    return BPE_OK;
}

extern "C" int bpe_dp_merge(bpe_ctx *c, int32_t iter) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(GK(c, k_dp_resolve), dim3(1), dim3(64), 0, c->stream, c->d_st, c->d_dp_key);
    LAUNCHCHK(c, "k_dp_resolve");
    c->dp_enq = iter + 1;
    if (c->slotted && c->slot2) {
        const uint32_t Z = 256u + (uint32_t)iter;
        if (c->dp_sparse) {
            CandArgs C;
            C.idx = c->d_idx;
            C.dirty = c->d_idx_dirty;
            C.cand = c->d_cand;
            C.stride = (uint32_t)c->idx_cap_words;
            C.T = (uint32_t)c->slot_T;
            C.enable = 1;
            C.tie_index = C.tie_window = 0;
            C.aa = aa_through_index(c) ? 1u : 0u;
            hipLaunchKernelGGL(GK(c, k_dp_cand), dim3(1), dim3(1024), 0, c->stream, c->d_st, C);
            LAUNCHCHK(c, "k_dp_cand");
        }
        c->dp_dl = delta_layout(c, Z);
        TRY(launch_passes2(c, Z, c->dp_sparse, c->dp_dl));
        hipLaunchKernelGGL(GK(c, k_dp_fold2), dim3((c->vcap + 255) / 256), dim3(256), 0, c->stream, c->d_delta, c->dp_dl, Z,
                           c->d_dp_folded, c->vcap, c->d_st);
        LAUNCHCHK(c, "k_dp_fold2");
        return BPE_OK;
    }
    if (c->slotted) return launch_merge_slot(c, 256u + (uint32_t)iter, iter, c->h_rec);
    const int saved = c->merge_impl;
    c->merge_impl = 0;  // the three-pass form finalises the pair before the rewrite
    const int rc = launch_merge(c, 256u + (uint32_t)iter, iter, c->h_rec, true);
    c->merge_impl = saved;
    return rc;
}

extern "C" int bpe_dp_apply(bpe_ctx *c, int32_t iter) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t Z = 256u + (uint32_t)iter;
    if (c->slotted && c->slot2) {
        const int mq0 = c->mq;
        hipLaunchKernelGGL(GK(c, k_dp_after_sum), dim3(1), dim3(1), 0, c->stream, c->d_st, c->d_dp_folded + 4 * (size_t)c->vcap);
        LAUNCHCHK(c, "k_dp_after_sum");
        TRY(launch_table2(c, Z, iter, c->h_rec, c->dp_sparse, c->dp_dl, true));
        if ((size_t)iter < c->dp_flip.size()) c->dp_flip[(size_t)iter] = (uint8_t)(c->mq != mq0);
        return BPE_OK;
    }
    if (c->slotted)  // the slotted pass leaves length/record bookkeeping to the table update
        TRY(launch_table_update<true>(c, c->d_dp_folded, Z, c->par ^ 1, c->h_rec, iter, 1));
    else
        TRY(launch_table_update<true>(c, c->d_dp_folded, Z, 0, nullptr, 0, 0));
    return BPE_OK;
}

// Wait for iteration `iter`'s record (written by the device into pinned memory).
extern "C" int bpe_dp_poll(bpe_ctx *c, int32_t iter, int32_t *a, int32_t *b, uint64_t *count,
                           uint64_t *local_len, int32_t *status) {
    if (!c || !c->d_dp_folded || iter < 0 || iter >= std::max(c->dp_merges, 1))
        return fail(c, BPE_E_ARG, "bad iteration");
    volatile IterRec *r = &c->h_rec[iter];
    for (uint64_t spins = 1; r->seq != (unsigned long long)iter + 1; spins++) {
        if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess &&
            r->seq != (unsigned long long)iter + 1)
            return fail(c, BPE_E_INTERNAL, "iteration %d never reported (stream idle)", iter);
    }
    __sync_synchronize();
    if (a) *a = r->a;
    if (b) *b = r->b;
    if (count) *count = r->count;
    if (local_len) *local_len = r->new_len;
    if (status) *status = (r->status == ST_OK) ? BPE_OK : (r->status == ST_EMPTY ? BPE_E_EMPTY_STATS : BPE_E_INTERNAL);
    if (r->status == ST_OK) {
        if (c->profile) c->prof_bytes[BPE_PROF_MERGE] += 4 * (2 * c->dp_cur_len + r->new_len);
        c->last_count = r->count;  // (global count: the same on every rank)
        if (r->a == r->b && c->idx_live && !aa_through_index(c)) c->idx_rebuild = true;
        c->dp_cur_len = r->new_len;
        c->n = r->new_len;  // tighter launch bound
        c->dp_done = iter + 1;
    }
    return BPE_OK;
}

extern "C" int bpe_dp_end(bpe_ctx *c) {
    if (!c) return BPE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->slotted) {
        // iterations enqueued after the last reported one (an early stop) did nothing on the
        // device: undo their parity flips, then hand back a contiguous stream
        if ((c->dp_enq - c->dp_done) & 1) c->par ^= 1;
        if (c->slot2) {
            for (int j = c->dp_done; j < c->dp_enq; j++)
                if ((size_t)j < c->dp_flip.size() && c->dp_flip[(size_t)j]) c->mq ^= 1;
        } else if ((c->dp_enq - c->dp_done) & 1) {
            c->mq ^= 1;
        }
        hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, c->stream, c->d_st, 0u);
        TRY(c->slot2 ? slots2_leave(c) : slots_leave(c));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n = c->dp_cur_len;
    } else if ((c->dp_enq - c->dp_done) & 1) {
        c->par ^= 1;
    }
    TRY(prof_drain(c));
    c->dp_nranks = 1;
    c->dp_rank = 0;
    c->dp_active = false;
    return BPE_OK;
}
