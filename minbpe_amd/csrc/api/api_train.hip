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
    TRY(ensure_srec(c));
    const bool delta = (c->mode == 1);
    // one event after every unit of work enqueued (an iteration, or a chain step of 1..CH_KMAX merges)
    EventList ev_list;  // destroyed on every exit path
    std::vector<hipEvent_t> &evs = ev_list.v;
    struct EvSpan {
        int ev, first, k;
    };
    std::vector<EvSpan> ev_spans;
    auto record_event = [&]() -> int {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        evs.push_back(e);
        if (hipEventRecord(e, c->stream) != hipSuccess) return -1;
        return (int)evs.size() - 1;
    };
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    HIPCHK(c, hipMemsetAsync(c->d_mat, 0, (size_t)c->vcap * c->vcap * sizeof(uint32_t), c->stream));
    // (a pass that a device status cut short in an earlier call may have left partial sums behind)
    HIPCHK(c, hipMemsetAsync(c->d_delta, 0, ((size_t)c->vcap * 4 * DELTA_REPL + 256 * DELTA_SKEW) * sizeof(uint32_t), c->stream));
    // (a second train() on this ctx: rows beyond 255 still hold the previous run's maxima, and the flag
    // words what its last table updates left)
    HIPCHK(c, hipMemsetAsync(c->d_rowmax, 0, (size_t)c->vcap * 2 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dbits, 0, DBITS_WORDS * sizeof(uint32_t), c->stream));
    TRY(prof_end(c));
    if (iter_ms_out && record_event() < 0) return fail(c, BPE_E_HIP, "hipEventRecord failed");
    // the byte stream and its statistics (iteration 0 of both modes): one pass over the bytes when it can be
    // (k_load_count), else widen -> chunk starts -> get_stats
    const bool fused_load = load_count_fusable(c);
    TRY(start_from_bytes(c, fused_load));
    const uint64_t n0 = c->n;
    if (!fused_load) TRY(launch_pair_count(c, false));

    int done = 0, rc = BPE_OK;
    uint64_t cur_len = n0;  // exact length after `done` merges
    bool stop = false;
    c->rep_shift = 5;
    const bool slots = delta && c->use_slots && c->merge_impl == 0;
    const bool form2 = slots && c->use_slots == 2;
    c->idx_live = false;
    c->idx_rebuild = false;
    c->last_count = ~0ull;
    c->n_sparse = c->n_dense = c->n_index_builds = 0;
    c->n_lean = c->n_deferred = 0;
    c->n_steps = c->n_full = c->n_chained = 0;
    c->n_fused = 0;
    static const char *stamp_path = getenv("BPE_STEP_STAMPS");
    const size_t stamp_bytes = (size_t)STEP_STAMP_RING * (3 * 16 + 256 * 2) * sizeof(unsigned long long);
    if (stamp_path) {
        if (!c->d_step_stamps) HIPCHK(c, hipMalloc((void **)&c->d_step_stamps, stamp_bytes));
        HIPCHK(c, hipMemsetAsync(c->d_step_stamps, 0, stamp_bytes, c->stream));
    }
    if (c->d_step_bar) {  // (a launch that a device status cut short in an earlier call may have left the count short)
        HIPCHK(c, hipMemsetAsync(c->d_step_bar, 0, 256, c->stream));
        c->step_bar_target = 0;
    }
    c->rows_pending = false;
    c->sum_valid = false;
    // The units of work in flight, oldest first.  GENERAL / LEAN: one iteration whose number the host knows.
    // CHAIN: a chain step (k_chain.hip) -- the device counts the merges, the host learns how many the step
    // did (0..CH_KMAX) from its step record.
    enum { U_GENERAL = 0, U_LEAN = 1, U_CHAIN = 2 };
    struct Unit {
        int kind;
        int iter;            // GENERAL / LEAN: the iteration; CHAIN: -1
        uint32_t step;       // CHAIN: its step record
        uint8_t hdr_flip;    // the launches flipped the header arrays (a sparse-style pass does not)
        uint8_t aa_indexed;  // GENERAL: its a == b pass keeps the index current by itself
        uint8_t pass_kind;   // LEAN / CHAIN: 1 = candidates from the index, 2 = every slot (statistics)
        int ev;              // event recorded after it (-1: none)
    };
    std::deque<Unit> q;
    int n_chain_inflight = 0;
    uint32_t steps = 0;  // chain steps enqueued so far (index of the next step record)
    // Iterations below general_until take the general path: the one a lean iteration / chain step handed back,
    // and -- a deferral costs a stream synchronisation, up to `depth` no-op units and the re-run (~200 us against
    // 25 us for a lean iteration and 44 us for a general one) -- the stretch after it when deferrals come
    // thick (streams whose late counts are 2 or 3: hundreds of tied pairs per selection, each of them a
    // deferral).  "Thick" = a tie the lean selection could not settle fewer than 16 lean merges after the last
    // such one, twice running: the stretch is then 32 iterations and doubles while that goes on, up to 1024;
    // a deferral that comes later ends it.  Pairs with a == b (the general path's merge by design) do not count.
    int general_until = -1, defer_hold = 0, defer_strikes = 0;
    uint64_t lean_at_last_defer = 0;
    bool lean_on = false;     // latched: the general path's kernels do not know a deferred iteration
    bool in_chain = false;    // the device counts the merges (k_set_iter ran, no GENERAL / LEAN unit enqueued since)
    bool records_ok = false;  // the last unit enqueued was a chain step: its table update's per-wave records are current
    c->ts = TILE2_MAX;  // (the early sweeps run on 1024-id slots; plan_pass2 re-packs into 256-id slots when the stream goes sparse)
    if (slots) TRY(form2 ? slots2_enter(c) : slots_enter(c));

    // ---- the record of merge j is final: outputs, statistics, what the next launches are sized by -------
    int deferred = -1;
    auto take_record = [&](int j, bool aa_indexed) -> int {
        volatile IterRec *r = &c->h_rec[j];
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
        if (r->a == r->b && c->idx_live && !aa_indexed) c->idx_rebuild = true;
        c->rep_shift = 5;  // (k_apply_delta folds 32 replicas with 16 loads in flight per lane: no need to shrink)
        done = j + 1;
        return BPE_OK;
    };
    // The device writes one IterRec per merge (and one StepRec per chain step) into pinned host memory; the
    // host runs up to `depth` units ahead and only ever waits on those records, never on the stream (no
    // hipStreamSynchronize in the loop).
    auto wait_iter = [&](int j) -> int {
        volatile IterRec *r = &c->h_rec[j];
        for (uint64_t spins = 1; r->seq != (unsigned long long)j + 1; spins++) {
            if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess &&
                r->seq != (unsigned long long)j + 1)
                return fail(c, BPE_E_INTERNAL, "iteration %d never reported (stream idle)", j);
        }
        __sync_synchronize();
        return BPE_OK;
    };
    auto wait_step = [&](uint32_t s) -> int {
        volatile StepRec *r = &c->h_srec[s % STEP_RING];
        for (uint64_t spins = 1; r->seq != (unsigned long long)s + 1; spins++) {
            if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess &&
                r->seq != (unsigned long long)s + 1)
                return fail(c, BPE_E_INTERNAL, "chain step %u never reported (stream idle)", s);
        }
        __sync_synchronize();
        return BPE_OK;
    };
    // A deferred unit: it and everything enqueued behind it merged nothing on the device (lean iterations
    // and chain steps only, all no-ops behind the deferred one); they did carry the stream length forward,
    // so the ping-pong parity (and any re-packing enqueued among them) stands as the host has it.  Take
    // their passes out of the statistics; merge `done` goes through the general path next.
    auto handle_deferral = [&](bool a_eq_b = false) -> int {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (const Unit &u : q) {
            if (u.kind == U_LEAN) {
                c->n_lean--;
                c->h_rec[u.iter].seq = 0;
            } else if (u.kind == U_CHAIN) {
                c->n_steps--;
            }
            if (u.pass_kind == 1) c->n_sparse--; else if (u.pass_kind == 2) c->n_dense--;
        }
        q.clear();
        n_chain_inflight = 0;
        // (the deferred unit's own selection launch re-scanned the rows flagged before it -- or left the flags
        // standing, for flush_lean_rows; the no-op table updates behind it left nothing new to re-scan)
        if (in_chain) {
            hipLaunchKernelGGL(GK(c, k_clear_defer_chain), dim3(1), dim3(1), 0, c->stream, c->d_st);
            c->rows_pending = true;  // (flag words may stand: k_rowmax_lean before the general selection)
        } else {
            c->rows_pending = false;
            hipLaunchKernelGGL(GK(c, k_clear_defer), dim3(1), dim3(1), 0, c->stream, c->d_st);
        }
        LAUNCHCHK(c, "k_clear_defer");
        c->n_deferred++;
        // back-off (see general_until above); nothing is in flight here, so the switch to the general path is safe,
        // and it runs until a lean iteration is enqueued again: the ordinary general -> lean hand-over
        if (c->lean_backoff && !a_eq_b) {
            defer_strikes = (c->n_lean - lean_at_last_defer < 16) ? defer_strikes + 1 : 0;
            lean_at_last_defer = c->n_lean;
            defer_hold = defer_strikes >= 2 ? std::min(32 << std::min(defer_strikes - 2, 5), 1024) : 0;
        } else if (a_eq_b) {
            defer_hold = 0;  // (one general iteration)
        }
        general_until = done + 1 + defer_hold;
        if (defer_hold) lean_on = false;
        in_chain = false;
        records_ok = false;
        c->sum_valid = false;
        deferred = -1;
        return BPE_OK;
    };

    c->repack_waste = 0.0;
    auto t_loop = now();
    while (!stop) {
        // ---- enqueue one more unit (if any merge is left to enqueue), then look at the oldest one ---------
        const int lb = done + (int)q.size();  // merges enqueued, at least (exact while no chain step is in flight)
        bool enqueued = false;
        if (lb < num_merges && (int)q.size() <= c->depth) {
            const int i = lb;
            bool full_rowmax = (i == 0);
            // Slots thinning out: re-pack (between merges nothing is pending).  A pass costs
            // per slot as much as per id, so the slot count should follow the stream length
            // closely; at 31/32 fill a whole cfg2 run re-packs ~45 times, ~60 us each.
            // (with the inverted index live most passes skip most slots, and a re-packing also costs
            // an index build: re-pack at 7/8 there)
            const uint64_t den = c->idx_live ? 8 : REPACK_DEN;
            bool repack = c->slotted && c->slot_T > 64 &&
                          c->n * den < c->slot_T * (uint64_t)(c->slot2 ? c->ts : TILE) * (den - 1);
            // (option repack_acc, dense phase: a sweep costs per SLOT, a re-pack about one sweep -- re-pack when the empty
            // fractions of the sweeps since the last one add up to what it costs, not at a fixed fill: early merges
            // remove 2 % of the stream each and 31/32 re-packs every other sweep)
            if (c->repack_acc && c->slotted && c->slot2 && !c->idx_live && c->slot_T > 64) {
                const double cap = (double)c->slot_T * (double)c->ts;
                c->repack_waste += std::max(0.0, 1.0 - (double)c->n / cap);
                repack = c->repack_waste * 100.0 >= (double)c->repack_acc;
            }
            bool sparse = false;
            // lean iterations (k_lean.hip) / chain steps (k_chain.hip): every sparse pass whose pair is rare enough,
            // and every pass of a stream too small for the index (a few thousand slots: visiting them all costs nothing)
            auto lean_wanted = [&](bool sp) {
                return c->slotted && c->slot2 && c->lean && i >= general_until &&
                       (lean_on || c->lean == 2 ||
                        (c->last_count != ~0ull && c->last_count <= (uint64_t)c->lean_count &&
                         (sp || c->slot_T <= 16 * SPARSE_GRID)));
            };
            // a GENERAL / LEAN unit needs its iteration number: not while chain steps are in flight (only a deferral
            // leads from chain steps to the general path, and it drains the queue)
            const bool known = n_chain_inflight == 0;
            if (known || in_chain) {
                c->vcur = 256u + (uint32_t)std::min(i, num_merges - 1);
                c->prof_iter = i;
                if (known && !delta && i > 0) {
                    TRY(clear_table(c));
                    TRY(launch_pair_count(c, false));
                    full_rowmax = true;
                }
                if (repack) {
                    c->repack_waste = 0.0;
                    if (c->slot2) {
                        TRY(slots2_leave(c));
                        TRY(slots2_enter(c));
                    } else {
                        TRY(slots_leave(c));
                        TRY(slots_enter(c));
                    }
                }
                if (c->slotted && c->slot2) TRY(plan_pass2(c, &sparse));
                const bool lean = lean_wanted(sparse);
                // dense chain steps: the early merges (no index yet, every id below LDSD_CAP), several pairs per sweep
                const int hi_dense = std::min(num_merges, done + (int)q.size() * CH_KDENSE + CH_KDENSE);
                const bool chain_dense = !lean && delta && c->chain && c->chain_dense && c->lean && c->lds_delta && c->slotted &&
                                         c->slot2 && !c->idx_live && !sparse && !full_rowmax && i >= general_until &&
                                         256 + hi_dense + 1 <= (int)CH_DCAP;
                const bool chain = chain_dense ||
                                   (lean && c->chain && c->lean_select && c->idx_live && c->tie_index && !full_rowmax);
                Unit u{U_GENERAL, i, 0u, 0, 0, 0, -1};
                if (chain) {
                    if (!chain_dense) lean_on = true;
                    if (!in_chain) {  // general iterations so far: the device counts from here
                        TRY(flush_lean_rows(c, c->vcur));  // (rows a lean table update left, if lean iterations ran before)
                        hipLaunchKernelGGL(GK(c, k_set_iter), dim3(1), dim3(1), 0, c->stream, c->d_st, (uint32_t)i, (uint32_t)num_merges);
                        LAUNCHCHK(c, "k_set_iter");
                        in_chain = true;
                        records_ok = false;
                    }
                    // the most this step can reach: every chain step in flight doing CH_KMAX merges
                    const int hi = chain_dense ? hi_dense : std::min(num_merges, done + (int)q.size() * CH_KMAX + CH_KMAX);
                    TRY(launch_chain_step(c, steps, 255u + (uint32_t)hi, sparse, records_ok, chain_dense));
                    u.kind = U_CHAIN;
                    u.iter = -1;
                    u.step = steps++;
                    u.pass_kind = sparse ? 1 : 2;
                    u.hdr_flip = chain_dense ? 1 : 0;
                    records_ok = true;
                    n_chain_inflight++;
                    enqueued = true;
                } else if (known) {
                    in_chain = false;
                    records_ok = false;
                    if (lean && !c->forced) {  // (a replay's selections are given: chain steps or the general path)
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
                        u.kind = U_LEAN;
                        u.pass_kind = sparse ? 1 : 2;
                        c->sum_valid = true;
                    } else {
                        c->sum_valid = false;
                        TRY(flush_lean_rows(c, c->vcur));
                        TRY(launch_select(c, full_rowmax, sparse));
                        if (c->slotted && c->slot2) {
                            const int mq0 = c->mq;
                            TRY(launch_merge2(c, 256u + (uint32_t)i, i, c->h_rec, sparse));
                            u.hdr_flip = (uint8_t)(c->mq != mq0);
                            u.aa_indexed = (uint8_t)c->last_aa_indexed;
                        } else if (c->slotted) {
                            TRY(launch_merge_slot(c, 256u + (uint32_t)i, i, c->h_rec));
                            u.hdr_flip = 1;
                        } else {
                            TRY(launch_merge(c, 256u + (uint32_t)i, i, c->h_rec, delta));
                        }
                    }
                    enqueued = true;
                }
                if (enqueued) {
                    if (iter_ms_out && (u.ev = record_event()) < 0) return fail(c, BPE_E_HIP, "hipEventRecord failed");
                    q.push_back(u);
                }
            }
        }
        // ---- consume the oldest unit once the host is `depth` ahead, or has nothing left to enqueue ---------
        if (!q.empty() && ((int)q.size() > c->depth || !enqueued)) {
            const Unit u = q.front();
            if (u.kind == U_CHAIN) {
                TRY(wait_step(u.step));
                const StepRec sr = *const_cast<const StepRec *>(&c->h_srec[u.step % STEP_RING]);
                if (sr.status == ST_DEFER) {
                    if ((int)sr.first_iter != done)
                        return fail(c, BPE_E_INTERNAL, "chain step %u deferred merge %u, the host expected %d", u.step, sr.first_iter, done);
                    TRY(handle_deferral((sr.pad >> 8) == 1));
                } else if (sr.status == ST_EMPTY) {
                    stop = true;
                    rc = fail(c, BPE_E_EMPTY_STATS, "max() arg is an empty sequence (iteration %d)", done);
                } else if (sr.status != ST_OK) {
                    stop = true;
                    rc = fail(c, BPE_E_INTERNAL, "device status %u at iteration %d (chain step %u)", sr.status, done, u.step);
                } else {
                    if (sr.k && (int)sr.first_iter != done)
                        return fail(c, BPE_E_INTERNAL, "chain step %u did merges from %u on, the host expected %d", u.step, sr.first_iter, done);
                    for (uint32_t j = 0; j < sr.k && !stop; j++) {
                        TRY(wait_iter((int)(sr.first_iter + j)));
                        TRY(take_record((int)(sr.first_iter + j), true));
                    }
                    if (sr.k) {
                        c->n_lean += sr.k;
                        if ((sr.pad & 0xFF) == CH_FULL) c->n_full++;
                        c->n_chained += (sr.pad & 0xFF) == CH_FULL ? sr.k - 1 : sr.k;
                        if (u.ev >= 0) ev_spans.push_back({u.ev, (int)sr.first_iter, (int)sr.k});
                    }
                    q.pop_front();
                    n_chain_inflight--;
                }
            } else {
                TRY(wait_iter(u.iter));
                TRY(take_record(u.iter, u.aa_indexed != 0));
                if (deferred >= 0) {
                    TRY(handle_deferral(c->h_rec[u.iter].a == c->h_rec[u.iter].b));
                } else if (!stop) {
                    if (u.ev >= 0) ev_spans.push_back({u.ev, u.iter, 1});
                    q.pop_front();
                }
            }
        }
        if (done >= num_merges) break;
    }
    c->prof_iter = -1;
    auto t_drain = now();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // (chain steps enqueued beyond the last merge did nothing but carry the stream length forward)
    if (!stop) {
        for (const Unit &u : q) {
            if (u.kind == U_CHAIN) c->n_steps--;
            if (u.pass_kind == 1) c->n_sparse--; else if (u.pass_kind == 2) c->n_dense--;
        }
        q.clear();
        TRY(flush_lean_rows(c, 256u + (uint32_t)done));
    }
    c->rows_pending = false;
    auto t_tail = now();
    if (c->slotted) {
        // leave the ids contiguous for whoever reads them next
        if (stop) {  // the failing unit and the no-op units enqueued after it: undo their parity flips
            const int back = (int)q.size();
            if (back & 1) c->par ^= 1;
            if (c->slot2) {
                for (const Unit &u : q)
                    if (u.hdr_flip) c->mq ^= 1;
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
        int prev = 0;  // (event 0: before the first pair count)
        for (const EvSpan &sp : ev_spans) {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, evs[(size_t)prev], evs[(size_t)sp.ev]));
            for (int j = 0; j < sp.k; j++)
                if (sp.first + j < done) iter_ms_out[sp.first + j] = ms / sp.k;
            prev = sp.ev;
        }
    }
    TRY(prof_drain(c));
    if (stamp_path && c->d_step_stamps) {  // (debug: [step % STEP_STAMP_RING][workgroup 0 | gm / 2 | last][16 stamps of the 100 MHz clock])
        std::vector<unsigned long long> h(stamp_bytes / sizeof(unsigned long long));
        HIPCHK(c, hipMemcpy(h.data(), c->d_step_stamps, stamp_bytes, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_path, "wb")) {
            fwrite(h.data(), 1, stamp_bytes, f);
            fclose(f);
        }
    }
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
