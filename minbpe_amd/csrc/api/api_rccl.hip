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

// The whole sharded training loop on the host side of the library: same protocol as
// minbpe_amd/dist.py (which remains the reference driver and the torch.distributed
// path), with the two per-merge all-reduces enqueued on the ctx's stream.
// len_out receives GLOBAL stream lengths (summed over ranks).
extern "C" int bpe_dp_train(bpe_ctx *c, int32_t num_merges, int32_t *pairs_out, uint64_t *counts_out,
                            uint64_t *len_out, int32_t *n_done) {
    if (!c || num_merges < 0) return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->comm) return fail(c, BPE_E_STATE, "bpe_comm_init first");
    RcclApi *r = rccl();
    if (n_done) *n_done = 0;
    TRY(bpe_dp_begin(c, num_merges, c->comm_rank, c->comm_nranks));
    RCCLCHK(c, r->AllReduce(c->d_dp_table, c->d_dp_table, 256 * 256, RCCL_INT32, RCCL_SUM, c->comm, c->stream));
    TRY(bpe_dp_table_ready(c));
    std::vector<long long> lens((size_t)std::max(num_merges, 1), 0);
    int consumed = 0, done = 0, rc = BPE_OK;
    bool stop = false;
    auto consume = [&](int j) -> int {
        int32_t a = 0, b = 0, status = 0;
        uint64_t cnt = 0, ll = 0;
        TRY(bpe_dp_poll(c, j, &a, &b, &cnt, &ll, &status));
        if (status != BPE_OK) {
            stop = true;
            rc = status == BPE_E_EMPTY_STATS
                     ? fail(c, BPE_E_EMPTY_STATS, "max() arg is an empty sequence (iteration %d)", j)
                     : fail(c, BPE_E_INTERNAL, "sharded training failed at iteration %d", j);
            return BPE_OK;
        }
        if (pairs_out) {
            pairs_out[2 * j] = a;
            pairs_out[2 * j + 1] = b;
        }
        if (counts_out) counts_out[j] = cnt;
        lens[(size_t)j] = (long long)ll;
        done = j + 1;
        return BPE_OK;
    };
    // Every rank issues the SAME collectives for every i in [0, num_merges), whatever it has learnt
    // about its own or a peer's failure in the meantime: once a device status is raised the kernels
    // of the remaining iterations are no-ops (each checks st->status), the all-reduces still match
    // up, and the third key word carries the status to every rank at the next merge (k_dp_key).
    // A host-side failure (a HIP or RCCL call that returns an error) cannot be recovered from --
    // the stream is lost -- and is reported after the peers have been released as far as possible.
    int host_rc = BPE_OK;
    auto keep = [&](int r_) {
        if (r_ != BPE_OK && host_rc == BPE_OK) host_rc = r_;
    };
    auto allreduce = [&](void *buf, size_t count, int dtype, int op) -> int {
        RCCLCHK(c, r->AllReduce(buf, buf, count, dtype, op, c->comm, c->stream));
        return BPE_OK;
    };
    for (int i = 0; i < num_merges; i++) {
        if (host_rc == BPE_OK) keep(bpe_dp_select(c, i));
        keep(allreduce(c->d_dp_key, 3, RCCL_INT64, RCCL_MIN));
        if (host_rc == BPE_OK) keep(bpe_dp_merge(c, i));
        keep(allreduce(c->d_dp_folded, (size_t)c->vcap * 4 + 64, RCCL_INT32, RCCL_SUM));
        if (host_rc == BPE_OK) keep(bpe_dp_apply(c, i));
        if (host_rc == BPE_OK && !stop && i - consumed >= c->depth) {
            keep(consume(consumed));
            if (!stop) consumed++;
        }
    }
    while (host_rc == BPE_OK && !stop && consumed < num_merges) {
        keep(consume(consumed));
        if (!stop) consumed++;
    }
    if (host_rc != BPE_OK) {
        c->dp_active = false;
        c->slotted = false;
        return host_rc;
    }
    TRY(bpe_dp_end(c));
    // global lengths: one SUM over the per-shard lengths
    if (num_merges > 0) {
        DevTmp t_l;
        HIPCHK(c, t_l.alloc((size_t)num_merges * 8));
        long long *d_l = t_l.as<long long>();
        HIPCHK(c, hipMemcpyAsync(d_l, lens.data(), (size_t)num_merges * 8, hipMemcpyHostToDevice, c->stream));
        RCCLCHK(c, r->AllReduce(d_l, d_l, (size_t)num_merges, RCCL_INT64, RCCL_SUM, c->comm, c->stream));
        HIPCHK(c, hipMemcpyAsync(lens.data(), d_l, (size_t)num_merges * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (len_out)
            for (int i = 0; i < done; i++) len_out[i] = (uint64_t)lens[(size_t)i];
    }
    if (n_done) *n_done = done;
    return rc;
}
