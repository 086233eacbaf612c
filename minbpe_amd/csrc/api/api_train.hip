// api_train.hip -- bpe_train: the pipelined training loop.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

extern "C" {

int bpe_train(bpe_ctx *c, int32_t num_merges, int32_t *pairs_out, uint64_t *counts_out,
              double *iter_ms_out, uint64_t *len_out, int32_t *n_done) {
    if (!c || num_merges < 0) return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->have_bytes) return fail(c, BPE_E_STATE, "bpe_load_bytes first");
    if (c->dp_active) return fail(c, BPE_E_STATE, "bpe_dp_end first");
    // BPE_TIMING=1: host wall time of the phases of this call on stderr (where does wall - device go?)
    static const bool timing = getenv("BPE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_begin = now();
    struct ProfIterReset {  // (whatever the exit path: the sampling of profiling events is this loop's business)
        bpe_ctx *c;
        ~ProfIterReset() { c->prof_iter = -1; }
    } prof_iter_reset{c};
    if (n_done) *n_done = 0;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_table(c, 256u + (uint32_t)num_merges));
    TRY(ensure_rec(c, std::max(num_merges, 1)));
    memset(c->h_rec, 0, sizeof(IterRec) * (size_t)std::max(num_merges, 1));
    TRY(start_from_bytes(c));
    const bool delta = (c->mode == 1);
    EventList ev_list;  // destroyed on every exit path
    std::vector<hipEvent_t> &evs = ev_list.v;
    if (iter_ms_out) {
        evs.assign((size_t)num_merges + 1, nullptr);
        for (auto &e : evs) HIPCHK(c, hipEventCreate(&e));
    }
    // statistics of the initial byte stream (iteration 0 of both modes)
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    HIPCHK(c, hipMemsetAsync(c->d_mat, 0, (size_t)c->vcap * c->vcap * sizeof(uint32_t), c->stream));
    // (a pass that a device status cut short in an earlier call may have left partial sums behind)
    HIPCHK(c, hipMemsetAsync(c->d_delta, 0, ((size_t)c->vcap * 4 * DELTA_REPL + 256 * DELTA_SKEW) * sizeof(uint32_t), c->stream));
    // (a second train() on this ctx: rows beyond 255 still hold the previous run's maxima, and the flag
    // words what its last table updates left)
    HIPCHK(c, hipMemsetAsync(c->d_rowmax, 0, (size_t)c->vcap * 2 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dbits, 0, DBITS_WORDS * sizeof(uint32_t), c->stream));
    TRY(prof_end(c));
    if (iter_ms_out) HIPCHK(c, hipEventRecord(evs[0], c->stream));
    const uint64_t n0 = c->n;
    TRY(launch_pair_count(c, false));

    int done = 0, rc = BPE_OK, consumed = 0;
    uint64_t cur_len = n0;  // exact length before iteration `consumed`
    bool stop = false;
    c->rep_shift = 5;
    const bool slots = delta && c->use_slots && c->merge_impl == 0;
    const bool form2 = slots && c->use_slots == 2;
    c->idx_live = false;
    c->idx_rebuild = false;
    c->last_count = ~0ull;
    c->n_sparse = c->n_dense = c->n_index_builds = 0;
    c->n_lean = c->n_deferred = 0;
    c->rows_pending = false;
    c->sum_valid = false;
    // lean iterations (k_lean.hip): which ones were enqueued that way (1: candidates from the index, 2: every
    // slot), the iteration that reported ST_DEFER, and the one iteration that must take the general path
    std::vector<uint8_t> lean_kind(form2 ? (size_t)num_merges : 0, 0);
    int deferred = -1;
    // Iterations below general_until take the general path: the one a lean iteration handed back, and -- a
    // deferral costs a stream synchronisation, up to `depth` no-op iterations and the re-run (~200 us against
    // 25 us for a lean iteration and 44 us for a general one) -- the stretch after it when deferrals come
    // thick (streams whose late counts are 2 or 3: hundreds of tied pairs per selection, each of them a
    // deferral).  The stretch doubles while they keep coming, up to 1024 iterations, and halves otherwise.
    int general_until = -1, defer_hold = 0, last_deferred_at = -(1 << 30);
    bool lean_on = false;  // latched: the general path's kernels do not know a deferred iteration
    // second form: which iterations flipped the header arrays (a sparse pass does not), so that an
    // early stop can undo the flips of the no-op iterations enqueued behind the failing one
    std::vector<uint8_t> hdr_flip(form2 ? (size_t)num_merges : 0, 0);
    // ... and which general-path iterations had an a == b pass that keeps the index current by itself (if the
    // pair turns out to have a == b, no rebuild is owed)
    std::vector<uint8_t> aa_indexed(form2 ? (size_t)num_merges : 0, 0);
    if (slots) TRY(form2 ? slots2_enter(c) : slots_enter(c));
    // The device writes one IterRec per iteration into pinned host memory; the
    // host runs up to `depth` iterations ahead and only ever waits on those
    // records, never on the stream (no hipStreamSynchronize in the loop).
    auto consume = [&](int j) -> int {
        volatile IterRec *r = &c->h_rec[j];
        for (uint64_t spins = 1; r->seq != (unsigned long long)j + 1; spins++) {
            if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess &&
                r->seq != (unsigned long long)j + 1)
                return fail(c, BPE_E_INTERNAL, "iteration %d never reported (stream idle)", j);
        }
        __sync_synchronize();
        if (r->status == ST_DEFER) {  // a == b: the lean path hands the merge back (handled by the loop below)
            deferred = j;
            return BPE_OK;
        }
        if (r->status == ST_EMPTY) {
            stop = true;
            rc = fail(c, BPE_E_EMPTY_STATS, "max() arg is an empty sequence (iteration %d)", j);
            return BPE_OK;
        }
        if (r->status != ST_OK) {
            stop = true;
            rc = fail(c, BPE_E_INTERNAL, "device status %u at iteration %d%s", r->status, j,
                      r->status == ST_LOOKBACK ? " (look-back wait timed out; set option merge=0)" : "");
            return BPE_OK;
        }
        if (pairs_out) {
            pairs_out[2 * j] = r->a;
            pairs_out[2 * j + 1] = r->b;
        }
        if (counts_out) counts_out[j] = r->count;
        if (len_out) len_out[j] = r->new_len;
        if (c->profile) {
            // algorithmic bytes: get_stats reads 4N_i, merge reads 4N_i, writes 4N_{i+1}
            if (delta) {
                c->prof_bytes[BPE_PROF_MERGE] += 4 * (2 * cur_len + r->new_len);
            } else {
                c->prof_bytes[BPE_PROF_MERGE] += 4 * (cur_len + r->new_len);
                if (j > 0) c->prof_bytes[BPE_PROF_PAIR_COUNT] += 4 * cur_len;
            }
        }
        cur_len = r->new_len;
        c->n = cur_len;  // tighter launch bound for what is enqueued next
        c->last_count = r->count;  // counts never grow: an upper bound for every later merge
        if (r->a == r->b && c->idx_live && !(form2 && aa_indexed[(size_t)j])) c->idx_rebuild = true;
        // sites per pass ~ the merged pair's count: fewer sites, fewer replicas to fold
        // few sites -> few same-address atomics -> fewer replicas to fold (measured: going
        // below 32 while a pass still has tens of thousands of sites slows the merge pass)
        c->rep_shift = 5;  // (k_apply_delta folds 32 replicas with 16 loads in flight per lane: no need to shrink)
        done = j + 1;
        return BPE_OK;
    };

    auto t_loop = now();
    int i = 0;
    while (!stop) {
        // enqueue iteration i (if any is left), then look at the record `depth` back
        if (i < num_merges) {
            c->vcur = 256u + (uint32_t)i;
            c->prof_iter = i;
            bool full_rowmax = (i == 0);
            if (!delta && i > 0) {
                TRY(clear_table(c));
                TRY(launch_pair_count(c, false));
                full_rowmax = true;
            }
            // Slots thinning out: re-pack (between merges nothing is pending).  A pass costs
            // per slot as much as per id, so the slot count should follow the stream length
            // closely; at 31/32 fill a whole cfg2 run re-packs ~45 times, ~60 us each.
            // (with the inverted index live most passes skip most slots, and a re-packing also costs
            // an index build: re-pack at 7/8 there)
            const uint64_t den = c->idx_live ? 8 : REPACK_DEN;
            if (c->slotted && c->slot_T > 64 && c->n * den < c->slot_T * (uint64_t)(c->slot2 ? TILE2 : TILE) * (den - 1)) {
                if (c->slot2) {
                    TRY(slots2_leave(c));
                    TRY(slots2_enter(c));
                } else {
                    TRY(slots_leave(c));
                    TRY(slots_enter(c));
                }
            }
            bool sparse = false;
            if (c->slotted && c->slot2) TRY(plan_pass2(c, &sparse));
            // lean iterations (k_lean.hip): every sparse pass whose pair is rare enough, and every pass of a
            // stream too small for the index (a few thousand slots: visiting them all costs nothing)
            const bool lean = c->slotted && c->slot2 && c->lean && i >= general_until &&
                              (lean_on || c->lean == 2 ||
                               (c->last_count != ~0ull && c->last_count <= (uint64_t)c->lean_count &&
                                (sparse || c->slot_T <= 16 * SPARSE_GRID)));
            if (lean) {
                lean_on = true;
                // (the selection works from the previous table update's records when that was a lean one
                // too: nothing else has touched the table or the row maxima since)
                if (c->lean_select && c->lean_sum && c->sum_valid && c->idx_live && c->tie_index && !full_rowmax) {
                    TRY(launch_sel_lean(c));
                } else if (c->lean_select && c->idx_live && c->tie_index && !full_rowmax) {
                    TRY(launch_rowsel_lean(c));
                } else {
                    TRY(flush_lean_rows(c, c->vcur));
                    TRY(launch_select(c, full_rowmax, false));
                }
                TRY(launch_lean(c, 256u + (uint32_t)i, i, c->h_rec, sparse));
                hdr_flip[(size_t)i] = 0;
                lean_kind[(size_t)i] = sparse ? 1 : 2;
                c->sum_valid = true;
            } else {
            c->sum_valid = false;
            TRY(flush_lean_rows(c, c->vcur));
            TRY(launch_select(c, full_rowmax, sparse));
            if (c->slotted && c->slot2) {
                const int mq0 = c->mq;
                TRY(launch_merge2(c, 256u + (uint32_t)i, i, c->h_rec, sparse));
                hdr_flip[(size_t)i] = (uint8_t)(c->mq != mq0);
                lean_kind[(size_t)i] = 0;
                aa_indexed[(size_t)i] = (uint8_t)c->last_aa_indexed;
            } else if (c->slotted)
                TRY(launch_merge_slot(c, 256u + (uint32_t)i, i, c->h_rec));
            else
                TRY(launch_merge(c, 256u + (uint32_t)i, i, c->h_rec, delta));
            }
            if (iter_ms_out) HIPCHK(c, hipEventRecord(evs[(size_t)i + 1], c->stream));
            i++;
        }
        if (consumed < i && (i - consumed > c->depth || i == num_merges)) {
            TRY(consume(consumed));
            if (deferred >= 0) {
                // Iterations [deferred, i) merged nothing on the device (all of them lean ones, all no-ops
                // behind the deferred one); they did carry the stream length forward, so the ping-pong
                // parity (and any re-packing enqueued among them) stands as the host has it.  Take their
                // passes out of the statistics, run iteration `deferred` through the general path and go
                // on from there.
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (int j = deferred; j < i; j++) {
                    c->n_lean--;
                    if (lean_kind[(size_t)j] == 1) c->n_sparse--; else c->n_dense--;
                    c->h_rec[j].seq = 0;
                }
                // (the deferred iteration's own selection launch re-scanned the rows of the merge before it;
                // the no-op table updates behind it left nothing to re-scan)
                c->rows_pending = false;
                hipLaunchKernelGGL(k_clear_defer, dim3(1), dim3(1), 0, c->stream, c->d_st);
                LAUNCHCHK(c, "k_clear_defer");
                c->n_deferred++;
                // back-off: a deferral within 32 lean iterations of the last one lengthens the general stretch
                // (nothing is in flight here, so the switch is safe; the general path then runs until a lean
                // iteration is enqueued again, which is the ordinary general -> lean hand-over)
                if (c->lean_backoff) {
                    if (deferred - last_deferred_at <= defer_hold + 32) defer_hold = std::min(std::max(2 * defer_hold, 32), 1024);
                    else defer_hold /= 2;
                }
                last_deferred_at = deferred;
                general_until = deferred + 1 + defer_hold;
                if (defer_hold) lean_on = false;
                i = deferred;
                deferred = -1;
            } else if (!stop) consumed++;
        }
        if (consumed >= num_merges) break;
    }
    c->prof_iter = -1;
    if (!stop) TRY(flush_lean_rows(c, 256u + (uint32_t)done));
    c->rows_pending = false;
    auto t_drain = now();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto t_tail = now();
    if (c->slotted) {
        // leave the ids contiguous for whoever reads them next
        if (stop) {  // parity of the no-op iterations enqueued after the failing one
            const int back = i - done;
            if (back & 1) c->par ^= 1;
            if (c->slot2) {
                for (int j = done; j < i; j++)
                    if (hdr_flip[(size_t)j]) c->mq ^= 1;
            } else if (back & 1) {
                c->mq ^= 1;
            }
            hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, c->stream, c->d_st, 0u);
        }
        TRY(c->slot2 ? slots2_leave(c) : slots_leave(c));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n = cur_len;
        c->vcur = 256u + (uint32_t)done;
    } else {
        c->par = done & 1;
        c->n = cur_len;
        c->vcur = 256u + (uint32_t)done;
    }
    // device buffers hold the stream after `done` merges
    if (iter_ms_out) {
        for (int i = 0; i < done; i++) {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, evs[(size_t)i], evs[(size_t)i + 1]));
            iter_ms_out[i] = ms;
        }
    }
    TRY(prof_drain(c));
    if (timing) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[bpe_train] %d merges: setup+enqueue of the first pass %.2f ms, loop %.2f ms, drain %.2f ms, "
                        "re-pack + timers %.2f ms\n", done, ms(t_begin, t_loop), ms(t_loop, t_drain), ms(t_drain, t_tail),
                ms(t_tail, now()));
    }
    if (n_done) *n_done = done;
    return rc;
}

}  // extern "C"
