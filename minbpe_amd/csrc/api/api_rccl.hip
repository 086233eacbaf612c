// api_rccl.hip -- librccl (dlopen), bpe_comm_*, bpe_dp_train.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// ---------------------------------------------------------------------------
// RCCL, straight from the library: the per-merge collectives are two tiny
// all-reduces, so the cost that matters is the host's enqueue path.  librccl is
// dlopen'ed (RTLD_LOCAL) so that a process that also runs torch.distributed
// keeps the two RCCL instances apart; the communicator lives on the ctx's stream.
namespace {
struct RcclUid { char internal[128]; };
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(RcclUid *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUid, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;  // (optional)
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int RCCL_INT32 = 2, RCCL_INT64 = 4, RCCL_SUM = 0, RCCL_MIN = 3;  // rccl.h: ncclDataType_t / ncclRedOp_t

RcclApi *rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.h) break;
        }
        if (api.h) {
            api.GetUniqueId = (int (*)(RcclUid *))dlsym(api.h, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void **, int, RcclUid, int))dlsym(api.h, "ncclCommInitRank");
            api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.h, "ncclAllReduce");
            api.CommDestroy = (int (*)(void *))dlsym(api.h, "ncclCommDestroy");
            api.CommAbort = (int (*)(void *))dlsym(api.h, "ncclCommAbort");
            api.GetErrorString = (const char *(*)(int))dlsym(api.h, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.h = nullptr;
        }
    }
    return api.h ? &api : nullptr;
}
#define RCCLCHK(c, call)                                                                     \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != 0)                                                                         \
            return fail((c), BPE_E_HIP, "%s failed: %s", #call,                              \
                        rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error"); \
    } while (0)
}  // namespace

extern "C" int bpe_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int bpe_comm_unique_id(uint8_t *out128) {
    if (!out128) return BPE_E_ARG;
    RcclApi *r = rccl();
    if (!r) return fail(nullptr, BPE_E_HIP, "librccl not found");
    RcclUid id;
    if (r->GetUniqueId(&id) != 0) return fail(nullptr, BPE_E_HIP, "ncclGetUniqueId failed");
    memcpy(out128, id.internal, 128);
    return BPE_OK;
}

extern "C" int bpe_comm_init(bpe_ctx *c, int32_t rank, int32_t nranks, const uint8_t *id128) {
    if (!c || !id128 || rank < 0 || nranks < 1 || rank >= nranks) return fail(c, BPE_E_ARG, "bad arguments");
    RcclApi *r = rccl();
    if (!r) return fail(c, BPE_E_HIP, "librccl not found");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) {
        r->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    RcclUid id;
    memcpy(id.internal, id128, 128);
    RCCLCHK(c, r->CommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return BPE_OK;
}

extern "C" int bpe_comm_destroy(bpe_ctx *c) {
    if (!c) return BPE_E_ARG;
    if (c->comm && rccl()) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        rccl()->CommDestroy(c->comm);
    }
    c->comm = nullptr;
    return BPE_OK;
}

// ---------------------------------------------------------------------------
// The sharded training loop: bpe_train's queue of units (api_train.hip) with the collectives between the launches.
// Two kinds of unit, chosen from GLOBAL facts only (the iteration number, the last merge's global count, what the
// step records said) so that every rank enqueues the same units -- and with them the same collectives -- in the same
// order, whatever its own shard looks like:
//   GENERAL  one merge, the five-launch iteration: bpe_dp_select -> MIN (3 x int64) -> bpe_dp_merge -> SUM
//            (4 vcap + 64 x int32) -> bpe_dp_apply.  The merges before the tied regime, and what a chain step hands
//            back (a == b at the head of the list, a tie some rank cannot order).
//   CHAIN    a chain step (k_chain.hip, launch_chain_step): 0..dp_kcap merges, MIN (98 x int64) + SUM (2 dp_kcap S + 64).
// Rank-local choices (re-packing, sparse or dense pass, index builds) change which kernels a rank runs, never what
// is exchanged.  The host runs `depth` units ahead and waits on records only, as on one GPU; a deferral drains the
// queue on every rank at the same unit (the step records are replicas).  After a device status every later unit is a
// no-op whose collectives still pair up.  A host-side failure (a HIP or comm call that returns an error) cannot be
// recovered from: the stream is lost; bpe_dp_train then aborts the library's communicator so that the peers fail
// instead of waiting (ncclCommAbort), a caller's communicator (bpe_dp_train_cb) is left to the caller's time-out.
namespace {
int dp_train_loop(bpe_ctx *c, int32_t num_merges, const DpComm &comm, int32_t *pairs_out, uint64_t *counts_out,
                  uint64_t *len_out, int32_t *n_done) {
    if (n_done) *n_done = 0;
    if (c->mode != 1 || c->use_slots != 2 || c->merge_impl != 0)
        return fail(c, BPE_E_STATE, "sharded training needs the default engine (mode 1, slots 2, merge 0)");
    struct CommScope {  // (launch_chain_step and dp_allreduce find the communicator in the ctx)
        bpe_ctx *c;
        ~CommScope() { c->dp_comm = nullptr; }
    } scope{c};
    c->dp_comm = &comm;
    TRY(bpe_dp_begin(c, num_merges, comm.rank, comm.nranks));
    TRY(ensure_srec(c));
    if (!c->d_dp_ckey) HIPCHK(c, hipMalloc((void **)&c->d_dp_ckey, DP_KEY_WORDS * sizeof(long long)));
    const uint64_t cfold_words = (uint64_t)2 * DP_KCAP_MAX * c->vcap + 64;
    if (cfold_words > c->cap_dp_cfold) {
        TRY(dev_realloc(c, c->d_dp_cfold, (size_t)cfold_words));
        c->cap_dp_cfold = cfold_words;
    }
    HIPCHK(c, hipMemsetAsync(c->d_dp_cfold, 0, cfold_words * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dp_ckey, 0, DP_KEY_WORDS * sizeof(long long), c->stream));
    TRY(dp_allreduce(c, c->d_dp_table, 2 * 256 * 256, BPE_DT_INT32, BPE_OP_SUM));  // (16-bit limbs: bpe_dp_begin)
    TRY(bpe_dp_table_ready(c));

    enum { U_GENERAL = 0, U_CHAIN = 2 };
    struct Unit {
        int kind, iter;
        uint32_t step;
        uint8_t hdr_flip, pass_kind;
    };
    std::deque<Unit> q;
    std::vector<long long> lens((size_t)std::max(num_merges, 1), 0);
    int done = 0, rc = BPE_OK, n_chain_inflight = 0;
    uint32_t steps = 0;
    bool stop = false, in_chain = false;
    int general_until = -1, defer_hold = 0, defer_strikes = 0;
    uint64_t lean_at_last_defer = 0, n_chain_merges = 0;
    c->n_lean = c->n_deferred = 0;
    c->n_steps = c->n_full = c->n_chained = 0;
    c->rows_pending = false;
    c->sum_valid = false;

    auto take_record = [&](int j) -> int {
        volatile IterRec *r = &c->h_rec[j];
        if (r->status == ST_EMPTY) {
            stop = true;
            rc = fail(c, BPE_E_EMPTY_STATS, "max() arg is an empty sequence (iteration %d)", j);
            return BPE_OK;
        }
        if (r->status != ST_OK) {
            stop = true;
            rc = fail(c, BPE_E_INTERNAL, "sharded training failed at iteration %d (device status %u)", j, r->status);
            return BPE_OK;
        }
        if (pairs_out) {
            pairs_out[2 * j] = r->a;
            pairs_out[2 * j + 1] = r->b;
        }
        if (counts_out) counts_out[j] = r->count;
        lens[(size_t)j] = (long long)r->new_len;
        if (c->profile) c->prof_bytes[BPE_PROF_MERGE] += 4 * (2 * c->dp_cur_len + r->new_len);
        c->dp_cur_len = r->new_len;
        c->n = r->new_len;         // (this shard's length: a tighter launch bound)
        c->last_count = r->count;  // (global: the same on every rank)
        if (r->a == r->b && c->idx_live && !aa_through_index(c)) c->idx_rebuild = true;
        done = j + 1;
        c->dp_done = done;
        return BPE_OK;
    };
    auto wait_iter = [&](int j) -> int {
        volatile IterRec *r = &c->h_rec[j];
        for (uint64_t spins = 1; r->seq != (unsigned long long)j + 1; spins++) {
            if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess && r->seq != (unsigned long long)j + 1)
                return fail(c, BPE_E_INTERNAL, "iteration %d never reported (stream idle)", j);
        }
        __sync_synchronize();
        return BPE_OK;
    };
    auto wait_step = [&](uint32_t s) -> int {
        volatile StepRec *r = &c->h_srec[s % STEP_RING];
        for (uint64_t spins = 1; r->seq != (unsigned long long)s + 1; spins++) {
            if ((spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) == hipSuccess && r->seq != (unsigned long long)s + 1)
                return fail(c, BPE_E_INTERNAL, "chain step %u never reported (stream idle)", s);
        }
        __sync_synchronize();
        return BPE_OK;
    };
    // a deferred chain step: it and the steps behind it merged nothing on any rank (api_train.hip, handle_deferral)
    auto handle_deferral = [&](bool a_eq_b) -> int {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (const Unit &u : q) {
            if (u.kind == U_CHAIN) c->n_steps--;
            if (u.pass_kind == 1) c->n_sparse--; else if (u.pass_kind == 2) c->n_dense--;
        }
        q.clear();
        n_chain_inflight = 0;
        hipLaunchKernelGGL(GK(c, k_clear_defer_chain), dim3(1), dim3(1), 0, c->stream, c->d_st);
        LAUNCHCHK(c, "k_clear_defer_chain");
        c->rows_pending = true;  // (flag words may stand: k_rowmax_lean before the general selection)
        c->n_deferred++;
        if (c->lean_backoff && !a_eq_b) {
            defer_strikes = (n_chain_merges - lean_at_last_defer < 16) ? defer_strikes + 1 : 0;
            lean_at_last_defer = n_chain_merges;
            defer_hold = defer_strikes >= 2 ? std::min(32 << std::min(defer_strikes - 2, 5), 1024) : 0;
        } else if (a_eq_b) {
            defer_hold = 0;
        }
        general_until = done + 1 + defer_hold;
        in_chain = false;
        return BPE_OK;
    };

    while (!stop && done < num_merges) {
        const int lb = done + (int)q.size();
        bool enqueued = false;
        if (lb < num_merges && (int)q.size() <= c->depth) {
            const int i = lb;
            // chain steps: from the first merge whose predecessor's GLOBAL count is small enough on (the same test
            // on every rank), outside the stretch a deferral handed to the general path
            const bool want_chain = c->chain && c->lean && c->lean_select && c->tie_index && i > 0 && i >= general_until &&
                                    c->last_count != ~0ull && c->last_count <= (uint64_t)c->lean_count;
            // dense chain steps before that regime: the early merges, several pairs per sweep over every slot (every
            // id below LDSD_CAP: the delta through LDS tables), ties left to the general path
            const int hi_dense = std::min(num_merges, done + (int)q.size() * CH_KDENSE + CH_KDENSE);
            const bool want_dense = !want_chain && c->chain && c->chain_dense && c->lean && c->lds_delta && i > 0 &&
                                    i >= general_until && c->last_count != ~0ull && 256 + hi_dense + 1 <= (int)CH_DCAP;
            const bool known = n_chain_inflight == 0;
            if (want_dense && (known || in_chain)) {
                c->vcur = 256u + (uint32_t)std::min(i, num_merges - 1);
                const uint64_t den = c->idx_live ? 8 : REPACK_DEN;
                if (c->slot_T > 64 && c->n * den < c->slot_T * (uint64_t)c->ts * (den - 1)) {
                    TRY(slots2_leave(c));
                    TRY(slots2_enter(c));
                }
                if (!in_chain) {
                    TRY(flush_lean_rows(c, c->vcur));
                    hipLaunchKernelGGL(GK(c, k_set_iter), dim3(1), dim3(1), 0, c->stream, c->d_st, (uint32_t)i, (uint32_t)num_merges);
                    LAUNCHCHK(c, "k_set_iter");
                    in_chain = true;
                }
                TRY(launch_chain_step(c, steps, 255u + (uint32_t)hi_dense, false, true, true));
                q.push_back(Unit{U_CHAIN, -1, steps++, 1, 2});
                n_chain_inflight++;
                enqueued = true;
            } else if (want_chain && (known || in_chain)) {
                c->vcur = 256u + (uint32_t)std::min(i, num_merges - 1);
                const uint64_t den = c->idx_live ? 8 : REPACK_DEN;
                if (c->slot_T > 64 && c->n * den < c->slot_T * (uint64_t)c->ts * (den - 1)) {
                    TRY(slots2_leave(c));
                    TRY(slots2_enter(c));
                }
                bool sparse = false;
                TRY(plan_pass2(c, &sparse));
                if (!c->idx_live || c->idx_rebuild) TRY(index_build(c));  // (a tie is ordered through the index, whatever the pass)
                if (!in_chain) {
                    TRY(flush_lean_rows(c, c->vcur));
                    hipLaunchKernelGGL(GK(c, k_set_iter), dim3(1), dim3(1), 0, c->stream, c->d_st, (uint32_t)i, (uint32_t)num_merges);
                    LAUNCHCHK(c, "k_set_iter");
                    in_chain = true;
                }
                const int hi = std::min(num_merges, done + (int)q.size() * c->dp_kcap + c->dp_kcap);
                TRY(launch_chain_step(c, steps, 255u + (uint32_t)hi, sparse, true, false));
                q.push_back(Unit{U_CHAIN, -1, steps++, 0, (uint8_t)(sparse ? 1 : 2)});
                n_chain_inflight++;
                enqueued = true;
            } else if (!want_chain && !want_dense && known) {
                in_chain = false;
                const int mq0 = c->mq;
                TRY(bpe_dp_select(c, i));
                TRY(dp_allreduce(c, c->d_dp_key, 3, BPE_DT_INT64, BPE_OP_MIN));
                TRY(bpe_dp_merge(c, i));
                TRY(dp_allreduce(c, c->d_dp_folded, (uint64_t)c->vcap * 4 + 64, BPE_DT_INT32, BPE_OP_SUM));
                TRY(bpe_dp_apply(c, i));
                q.push_back(Unit{U_GENERAL, i, 0u, (uint8_t)(c->mq != mq0), 0});
                enqueued = true;
            }
        }
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
                    rc = fail(c, BPE_E_INTERNAL, "sharded training failed at iteration %d (device status %u, chain step %u)", done,
                              sr.status, u.step);
                } else {
                    if (sr.k && (int)sr.first_iter != done)
                        return fail(c, BPE_E_INTERNAL, "chain step %u did merges from %u on, the host expected %d", u.step, sr.first_iter, done);
                    for (uint32_t j = 0; j < sr.k && !stop; j++) {
                        TRY(wait_iter((int)(sr.first_iter + j)));
                        TRY(take_record((int)(sr.first_iter + j)));
                    }
                    if (sr.k) {
                        c->n_lean += sr.k;
                        n_chain_merges += sr.k;
                        if ((sr.pad & 0xFF) == CH_FULL) c->n_full++;
                        c->n_chained += (sr.pad & 0xFF) == CH_FULL ? sr.k - 1 : sr.k;
                    }
                    q.pop_front();
                    n_chain_inflight--;
                }
            } else {
                TRY(wait_iter(u.iter));
                TRY(take_record(u.iter));
                if (!stop) q.pop_front();
            }
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!stop) {
        for (const Unit &u : q) {
            if (u.kind == U_CHAIN) c->n_steps--;
            if (u.pass_kind == 1) c->n_sparse--; else if (u.pass_kind == 2) c->n_dense--;
        }
        q.clear();
        TRY(flush_lean_rows(c, 256u + (uint32_t)done));
    }
    c->rows_pending = false;
    // leave the ids contiguous; the failing unit and the no-op units behind it: undo their parity flips
    if (stop) {
        if (q.size() & 1) c->par ^= 1;
        for (const Unit &u : q)
            if (u.hdr_flip) c->mq ^= 1;
        hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, c->stream, c->d_st, 0u);
    }
    TRY(slots2_leave(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n = c->dp_cur_len;
    c->vcur = 256u + (uint32_t)done;
    TRY(prof_drain(c));
    c->dp_nranks = 1;
    c->dp_rank = 0;
    c->dp_active = false;
    // global lengths: one SUM over the per-shard lengths (every rank, also after a failure: `done` is the same everywhere)
    if (num_merges > 0) {
        DevTmp t_l;
        HIPCHK(c, t_l.alloc((size_t)num_merges * 8));
        long long *d_l = t_l.as<long long>();
        HIPCHK(c, hipMemcpyAsync(d_l, lens.data(), (size_t)num_merges * 8, hipMemcpyHostToDevice, c->stream));
        TRY(dp_allreduce(c, d_l, (uint64_t)num_merges, BPE_DT_INT64, BPE_OP_SUM));
        HIPCHK(c, hipMemcpyAsync(lens.data(), d_l, (size_t)num_merges * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (len_out)
            for (int i = 0; i < done; i++) len_out[i] = (uint64_t)lens[(size_t)i];
    }
    if (n_done) *n_done = done;
    return rc;
}

int rccl_allreduce(void *user, void *buf, uint64_t count, int32_t dtype, int32_t op, void *stream) {
    bpe_ctx *c = (bpe_ctx *)user;
    return rccl()->AllReduce(buf, buf, (size_t)count, dtype == BPE_DT_INT64 ? RCCL_INT64 : RCCL_INT32,
                             op == BPE_OP_MIN ? RCCL_MIN : RCCL_SUM, c->comm, (hipStream_t)stream);
}
}  // namespace

// librccl on the ctx's stream: the collectives are enqueued like kernels, the host never waits for one.
extern "C" int bpe_dp_train(bpe_ctx *c, int32_t num_merges, int32_t *pairs_out, uint64_t *counts_out,
                            uint64_t *len_out, int32_t *n_done) {
    if (!c || num_merges < 0) return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->comm || !rccl()) return fail(c, BPE_E_STATE, "bpe_comm_init first");
    DpComm comm;
    comm.fn = rccl_allreduce;
    comm.user = c;
    comm.rank = c->comm_rank;
    comm.nranks = c->comm_nranks;
    const int rc = dp_train_loop(c, num_merges, comm, pairs_out, counts_out, len_out, n_done);
    // A device status stops every rank at the same merge (it travels with the collectives).  A HOST-side failure on this
    // rank alone -- a HIP or RCCL call that returned an error, a unit that never reported -- leaves the peers inside a
    // collective this rank will never join: abort the communicator, so that they fail with an RCCL error instead of
    // waiting for ever (the communicator is gone afterwards: bpe_comm_init again).  With bpe_dp_train_cb the
    // collectives are the caller's, and so is their time-out (torch.distributed: the process group's).
    if ((rc == BPE_E_HIP || rc == BPE_E_INTERNAL) && c->comm && rccl()->CommAbort) {
        (void)rccl()->CommAbort(c->comm);
        c->comm = nullptr;
    }
    return rc;
}

// the caller's all-reduce (torch.distributed through a ctypes callback, a test's host-side reduction)
extern "C" int bpe_dp_train_cb(bpe_ctx *c, int32_t num_merges, int32_t rank, int32_t nranks, bpe_allreduce_fn fn, void *user,
                               int32_t *pairs_out, uint64_t *counts_out, uint64_t *len_out, int32_t *n_done) {
    if (!c || num_merges < 0 || !fn || rank < 0 || nranks < 1 || rank >= nranks) return fail(c, BPE_E_ARG, "bad arguments");
    DpComm comm;
    comm.fn = fn;
    comm.user = user;
    comm.rank = rank;
    comm.nranks = nranks;
    return dp_train_loop(c, num_merges, comm, pairs_out, counts_out, len_out, n_done);
}
