// bpe_api.hip -- host side of the C-ABI declared in include/bpe_hip.h.
//
// One ctx per GPU.  All work of a ctx is queued on one HIP stream; the id
// stream, the pair table and all scratch live in HBM for the lifetime of the
// ctx.  The training loop is host-sequenced (the
// reference's outer loop is inherently sequential, basic.py:31) but every
// iteration runs on the device with no id data crossing PCIe.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bpe_hip.h"
#include "bpe_kernels.hip"

using namespace bpe;

#ifndef REPACK_DEN
#define REPACK_DEN 32  // re-pack the slots when their fill drops below (REPACK_DEN-1)/REPACK_DEN
#endif

namespace {
thread_local std::string g_create_err;

struct ProfEv {
    int kind;
    hipEvent_t e0, e1;
    uint64_t bytes;
};
}  // namespace

struct bpe_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // resident input (bpe_load_bytes)
    uint8_t *d_bytes = nullptr;
    uint64_t nbytes = 0, cap_bytes = 0;
    uint64_t *d_offsets = nullptr;
    uint64_t n_chunks = 0, cap_offsets = 0;
    bool have_bytes = false;
    uint8_t *d_wexp = nullptr;  // per chunk: weight exponent (bpe_load_bytes_weighted)
    uint64_t cap_wexp = 0;
    bool weighted = false;

    // id stream
    uint32_t *d_ids[2] = {nullptr, nullptr};
    uint64_t cap_ids = 0;
    int par = 0;
    uint64_t n = 0;
    bool have_ids = false;

    // pair table
    uint32_t *d_mat = nullptr, *d_first = nullptr, *d_rowmax = nullptr;
    uint32_t vcap = 0;  // matrix dimension == row stride
    uint32_t vcur = 0;  // ids in use: [0, vcur)
    bool stats_valid = false;

    DevState *d_st = nullptr;
    uint64_t *d_tsum = nullptr, *d_tile_off = nullptr;
    uint8_t *d_tile_sin = nullptr;
    uint64_t cap_tiles = 0;
    IterRec *h_rec = nullptr;  // pinned, device-visible
    int rec_cap = 0;
    unsigned long long *d_scratch = nullptr;  // 2 x u64 cursor/counter
    uint32_t *d_delta = nullptr;       // 4 x vcap: decL | decR | incL | incR
    uint32_t *d_dirty_list = nullptr;  // rows whose rowmax must be recomputed
    uint32_t *d_dirty_n = nullptr;
    int depth = 8;  // iterations the host may run ahead of the device
    // slotted stream (training loop, a != b merges)
    int use_slots = 1;
    int fused_rows = 0;                  // 1: row maxima inside the k_apply_delta launch
    uint32_t sel_epoch = 0;              // k_select decision flag value of the last launch
    unsigned long long apply_target = 0;  // apply blocks launched since the state was initialised
    int rep_shift = 5;        // log2(delta-vector replicas in use): shrinks as merges get rarer
    bool slotted = false;
    uint64_t slot_T = 0;
    int mq = 0;
    uint32_t *d_meta[2] = {nullptr, nullptr};
    uint4 *d_hdr[2] = {nullptr, nullptr};  // per slot: first three words, last word
    uint32_t *d_slot_lens = nullptr;
    unsigned long long *d_slot_off = nullptr, *d_slot_bsum = nullptr;
    uint32_t *d_ids2 = nullptr;  // third stream buffer: target of compactions
    uint64_t cap_slots = 0;
    // data-parallel stepping (bpe_dp_*)
    int dp_rank = 0, dp_nranks = 1, dp_merges = 0, dp_enq = 0, dp_done = 0;
    bool dp_active = false;  // between bpe_dp_begin and bpe_dp_end
    void *comm = nullptr;    // RCCL communicator (bpe_comm_init), one rank per ctx
    int comm_rank = 0, comm_nranks = 1;
    uint32_t *d_dp_folded = nullptr, *d_dp_table = nullptr;
    long long *d_dp_key = nullptr;
    uint64_t dp_cur_len = 0;
    int merge_impl = 0;  // 0 three-pass | 1 single-pass (two-level decoupled look-back)
    unsigned long long *d_desc = nullptr;   // look-back descriptors, one per tile
    unsigned long long *d_gdesc = nullptr;  // ... and one per group of 64 tiles
    uint64_t cap_desc = 0;
    uint32_t epoch = 0;
    uint32_t lb_tune = 1;  // bits 0..7: s_sleep(8) units between polls; bit 8: measurement-only 'no wait'

    // encode scratch (grow-only)
    uint32_t *d_enc_tmp = nullptr, *d_enc_len = nullptr;
    int32_t *d_enc_out = nullptr;
    unsigned long long *d_enc_off = nullptr, *d_enc_bsum = nullptr, *d_enc_long = nullptr;
    unsigned long long *d_ht_keys = nullptr;
    uint32_t *d_ht_vals = nullptr;
    int32_t *d_merge_ids = nullptr;
    uint64_t cap_enc_n = 0, cap_enc_chunks = 0, cap_ht = 0, cap_merge_ids = 0;

    // decode (grow-only): vocab table, then ids / lengths / offsets / bytes of the last batch
    uint8_t *d_dec_blob = nullptr, *d_dec_out = nullptr;
    unsigned long long *d_dec_voff = nullptr, *d_dec_off = nullptr, *d_dec_bsum = nullptr;
    int32_t *d_dec_ids = nullptr;
    uint32_t *d_dec_len = nullptr;
    uint64_t cap_dec_blob = 0, cap_dec_voff = 0, cap_dec_n = 0, cap_dec_out = 0;
    uint32_t dec_V = 0;
    bool dec_have_vocab = false, dec_have_result = false;
    uint64_t dec_n = 0, dec_total = 0;

    int mode = 1;     // 0 recount | 1 delta
    int profile = 0;  // 0 off | 1 hipEvents around the merge pass | 2 around every kernel class
    bool prof_active = false;
    int k1 = 2;       // 0 one atomic per position | 1 LDS hash cache | 2 = 1 + dense 16-bit LDS table for byte streams
    bool stream_is_bytes = false;  // every id of the current stream is < 256 (fresh from k_widen)

    std::vector<ProfEv> prof_open;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[BPE_PROF_NKINDS] = {0};
    uint64_t prof_launches[BPE_PROF_NKINDS] = {0};
    uint64_t prof_bytes[BPE_PROF_NKINDS] = {0};
};

namespace {

int fail(bpe_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail((c), BPE_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                             \
    } while (0)

#define LAUNCHCHK(c, name)                                                                \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return fail((c), BPE_E_HIP, "launch %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != BPE_OK) return rc_; \
    } while (0)

// scratch device allocation of one call, released on every exit path
struct DevTmp {
    void *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T>
    T *as() const { return (T *)p; }
};

struct EventList {
    std::vector<hipEvent_t> v;
    ~EventList() {
        for (hipEvent_t e : v)
            if (e) (void)hipEventDestroy(e);
    }
};

template <typename T>
int dev_realloc(bpe_ctx *c, T *&p, size_t count) {
    if (p) HIPCHK(c, hipFree(p));
    p = nullptr;
    if (count) HIPCHK(c, hipMalloc((void **)&p, count * sizeof(T)));
    return BPE_OK;
}

inline uint64_t ntiles_of(uint64_t n) { return (n + TILE - 1) / TILE; }

int ensure_ids(bpe_ctx *c, uint64_t n) {
    // capacity padded so every tile load (and the +1 halo word) is in bounds
    const uint64_t need = (ntiles_of(n) + 2) * TILE;
    if (need > c->cap_ids) {
        TRY(dev_realloc(c, c->d_ids[0], need));
        TRY(dev_realloc(c, c->d_ids[1], need));
        TRY(dev_realloc(c, c->d_ids2, need));
        c->cap_ids = need;
    }
    const uint64_t nt = ntiles_of(n) + 1;
    if (nt > c->cap_tiles) {
        TRY(dev_realloc(c, c->d_tsum, nt));
        TRY(dev_realloc(c, c->d_tile_off, nt));
        TRY(dev_realloc(c, c->d_tile_sin, nt));
        TRY(dev_realloc(c, c->d_meta[0], nt));
        TRY(dev_realloc(c, c->d_meta[1], nt));
        TRY(dev_realloc(c, c->d_hdr[0], nt));
        TRY(dev_realloc(c, c->d_hdr[1], nt));
        TRY(dev_realloc(c, c->d_slot_lens, nt));
        TRY(dev_realloc(c, c->d_slot_off, nt + 1));
        TRY(dev_realloc(c, c->d_slot_bsum, nt / SCAN_TILE + 2));
        TRY(dev_realloc(c, c->d_desc, nt));
        TRY(dev_realloc(c, c->d_gdesc, nt / 64 + 2));
        HIPCHK(c, hipMemsetAsync(c->d_desc, 0, nt * sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_gdesc, 0, (nt / 64 + 2) * sizeof(unsigned long long), c->stream));
        c->cap_tiles = nt;
    }
    return BPE_OK;
}

int ensure_table(bpe_ctx *c, uint32_t v) {
    if (v > 65535) return fail(c, BPE_E_LIMIT, "vocab %u exceeds this build's 65535 limit", v);
    if (v <= c->vcap) return BPE_OK;
    uint32_t nv = std::max<uint32_t>(v, 256);
    nv = (nv + 63) & ~63u;  // rows stay 256 B aligned
    TRY(dev_realloc(c, c->d_mat, (size_t)nv * nv));
    TRY(dev_realloc(c, c->d_rowmax, (size_t)nv));
    TRY(dev_realloc(c, c->d_delta, (size_t)nv * 4 * DELTA_REPL));
    TRY(dev_realloc(c, c->d_dirty_list, (size_t)nv));
    if (!c->d_dirty_n) HIPCHK(c, hipMalloc((void **)&c->d_dirty_n, sizeof(uint32_t)));
    HIPCHK(c, hipMemsetAsync(c->d_delta, 0, (size_t)nv * 4 * DELTA_REPL * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dirty_n, 0, sizeof(uint32_t), c->stream));
    if (c->d_first) {
        HIPCHK(c, hipFree(c->d_first));
        c->d_first = nullptr;
    }
    c->vcap = nv;
    c->stats_valid = false;
    return BPE_OK;
}

int ensure_rec(bpe_ctx *c, int n) {
    if (n <= c->rec_cap) return BPE_OK;
    if (c->h_rec) HIPCHK(c, hipHostFree(c->h_rec));
    c->h_rec = nullptr;
    HIPCHK(c, hipHostMalloc((void **)&c->h_rec, sizeof(IterRec) * (size_t)n, hipHostMallocMapped));
    c->rec_cap = n;
    return BPE_OK;
}

// ---- profiling --------------------------------------------------------------
int prof_begin(bpe_ctx *c, int kind, uint64_t bytes) {
    // level 1: only the dominant kernel class (merge) -- two event records per
    // iteration; level 2: every class (adds marker packets between all kernels)
    c->prof_active = c->profile >= 2 || (c->profile == 1 && kind == BPE_PROF_MERGE);
    if (!c->prof_active) return BPE_OK;
    ProfEv ev;
    ev.kind = kind;
    ev.bytes = bytes;
    for (hipEvent_t *e : {&ev.e0, &ev.e1}) {
        if (!c->ev_pool.empty()) {
            *e = c->ev_pool.back();
            c->ev_pool.pop_back();
        } else {
            HIPCHK(c, hipEventCreate(e));
        }
    }
    HIPCHK(c, hipEventRecord(ev.e0, c->stream));
    c->prof_open.push_back(ev);
    return BPE_OK;
}
int prof_end(bpe_ctx *c) {
    if (!c->prof_active) return BPE_OK;
    c->prof_active = false;
    HIPCHK(c, hipEventRecord(c->prof_open.back().e1, c->stream));
    return BPE_OK;
}
int prof_drain(bpe_ctx *c) {
    if (c->prof_open.empty()) return BPE_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (ProfEv &ev : c->prof_open) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1));
        c->prof_ms[ev.kind] += ms;
        c->prof_launches[ev.kind] += 1;
        c->prof_bytes[ev.kind] += ev.bytes;
        c->ev_pool.push_back(ev.e0);
        c->ev_pool.push_back(ev.e1);
    }
    c->prof_open.clear();
    return BPE_OK;
}

// ---- launch helpers -----------------------------------------------------------
inline unsigned grid_for(uint64_t work_items, unsigned per_block, unsigned cap) {
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// widen resident bytes into ids[0], mark chunk starts, reset state
int start_from_bytes(bpe_ctx *c) {
    const uint64_t n = c->nbytes;
    TRY(ensure_ids(c, n));
    TRY(prof_begin(c, BPE_PROF_WIDEN, 5 * n));
    if (n) {
        hipLaunchKernelGGL(k_widen, dim3(grid_for(n, 256 * 16, c->num_cus * 8)), dim3(256), 0,
                           c->stream, c->d_bytes, c->d_ids[0], n);
        LAUNCHCHK(c, "k_widen");
        if (c->n_chunks) {
            hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(c->n_chunks, 256, c->num_cus * 8)),
                               dim3(256), 0, c->stream, c->d_ids[0], c->d_offsets, c->n_chunks, n);
            LAUNCHCHK(c, "k_mark_starts");
            if (c->weighted) {
                hipLaunchKernelGGL(k_mark_weights, dim3(grid_for(c->n_chunks, 256, c->num_cus * 8)), dim3(256),
                                   0, c->stream, c->d_ids[0], c->d_offsets, c->d_wexp, c->n_chunks, n);
                LAUNCHCHK(c, "k_mark_weights");
            }
        }
    }
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1), 0, c->stream, c->d_st, (unsigned long long)n);
    LAUNCHCHK(c, "k_init_state");
    c->apply_target = 0;
    TRY(prof_end(c));
    c->par = 0;
    c->n = n;
    c->vcur = 256;
    c->have_ids = true;
    c->stats_valid = false;
    c->stream_is_bytes = true;
    return BPE_OK;
}

int clear_table(bpe_ctx *c) {
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    HIPCHK(c, hipMemsetAsync(c->d_mat, 0, (size_t)c->vcur * c->vcap * sizeof(uint32_t), c->stream));
    TRY(prof_end(c));
    return BPE_OK;
}

// K1 on the current stream into the (cleared) table
int launch_pair_count(bpe_ctx *c, bool with_first) {
    const uint64_t n = c->n;
    TRY(prof_begin(c, BPE_PROF_PAIR_COUNT, 4 * n));
    if (n >= 2) {
        if (with_first) {
            hipLaunchKernelGGL(k_pair_count_simple<true>, dim3(grid_for(n, 1024, c->num_cus * 8)),
                               dim3(256), 0, c->stream, c->d_ids[c->par], c->d_st, c->par, c->d_mat,
                               c->vcap, c->d_first);
        } else if (c->k1 == 0) {
            hipLaunchKernelGGL(k_pair_count_simple<false>, dim3(grid_for(n, 1024, c->num_cus * 8)),
                               dim3(256), 0, c->stream, c->d_ids[c->par], c->d_st, c->par, c->d_mat,
                               c->vcap, (uint32_t *)nullptr);
        } else if (c->k1 == 2 && c->vcur <= 256 && c->stream_is_bytes && !c->weighted) {
            // (16-bit LDS counters: unit increments only)
            hipLaunchKernelGGL(k_pair_count_bytes, dim3(grid_for(n, 4 * PC_THREADS, c->num_cus)),
                               dim3(PC_THREADS), PC_LDS_BYTES, c->stream, c->d_ids[c->par], c->d_st,
                               c->par, c->d_mat, c->vcap);
        } else {
            hipLaunchKernelGGL(k_pair_count_lds, dim3(grid_for(n, 4 * PC_THREADS, c->num_cus)),
                               dim3(PC_THREADS), PC_LDS_BYTES, c->stream, c->d_ids[c->par], c->d_st,
                               c->par, c->d_mat, c->vcap);
        }
        LAUNCHCHK(c, "k_pair_count");
    }
    TRY(prof_end(c));
    return BPE_OK;
}

inline uint32_t vcap_rep(const bpe_ctx *c) { return c->vcap | ((uint32_t)c->rep_shift << 24); }

SlotRef stream_ref(const bpe_ctx *c) {
    SlotRef r;
    if (c->slotted) {
        r.b0 = c->d_ids[0];
        r.b1 = c->d_ids[1];
        r.meta = c->d_meta[c->mq];
        r.T = c->slot_T;
    } else {
        r.b0 = c->d_ids[c->par];
        r.b1 = nullptr;
        r.meta = nullptr;
        r.T = 0;
    }
    return r;
}

// K2 + tie-break: after these, resolved_pair() gives the pair on the device
int launch_select(bpe_ctx *c, bool rowmax_all) {
    TRY(prof_begin(c, BPE_PROF_ARGMAX, 0));
    if (rowmax_all) {
        hipLaunchKernelGGL(k_rowmax_all, dim3(c->vcur), dim3(256), 0, c->stream, c->d_mat, c->vcap,
                           c->vcur, c->d_rowmax);
        LAUNCHCHK(c, "k_rowmax_all");
    }
    const SlotRef ref = stream_ref(c);
    const uint64_t space = c->slotted ? c->slot_T * TILE : c->n;
    const unsigned blocks = space > TIE_WINDOW0 ? grid_for(space - TIE_WINDOW0, 1024, TIE_BLOCKS) : 1u;
    hipLaunchKernelGGL(k_select, dim3(blocks), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                       c->vcap, c->vcur, c->d_st, ref, c->par, c->dp_active ? 1 : 0, ++c->sel_epoch);
    LAUNCHCHK(c, "k_select");
    TRY(prof_end(c));
    return BPE_OK;
}

// table update: apply blocks + row-maxima blocks in one launch
template <bool FOLDED>
int launch_table_update(bpe_ctx *c, uint32_t *delta, uint32_t Z, int par, IterRec *rec, int iter,
                        int slot_finish) {
    const uint32_t na = (Z + 1 + 31) / 32;
    // Measured (cfg2): handing the row maxima to extra blocks of the same launch costs more
    // (release + acquire fences, polling) than the ~1.5 us kernel boundary it saves -- 58 vs
    // 45 ms per train -- so by default they are a launch of their own.
    if (c->fused_rows) {
        c->apply_target += na;
        hipLaunchKernelGGL(k_apply_delta<FOLDED>, dim3(na + ROW_BLOCKS), dim3(256), 0, c->stream, c->d_mat,
                           c->vcap, delta, FOLDED ? c->vcap : vcap_rep(c), c->d_rowmax, c->d_st, Z,
                           c->d_dirty_list, c->d_dirty_n, par, rec, iter, slot_finish, na,
                           c->apply_target);
    } else {
        hipLaunchKernelGGL(k_apply_delta<FOLDED>, dim3(na), dim3(256), 0, c->stream, c->d_mat, c->vcap,
                           delta, FOLDED ? c->vcap : vcap_rep(c), c->d_rowmax, c->d_st, Z,
                           c->d_dirty_list, c->d_dirty_n, par, rec, iter, slot_finish, na, 0ull);
        LAUNCHCHK(c, "k_apply_delta");
        hipLaunchKernelGGL(k_rowmax_list, dim3(ROW_BLOCKS), dim3(256), 0, c->stream, c->d_mat, c->vcap,
                           Z + 1, c->d_rowmax, c->d_st, c->d_dirty_list, c->d_dirty_n);
    }
    LAUNCHCHK(c, "k_apply_delta");
    return BPE_OK;
}

// K3: three passes (summary, tile scan, rewrite); flips the ping-pong parity.
// with_delta: the rewrite pass also accumulates the pair-table delta vectors,
// which k_apply_delta / k_rowmax_list then fold into the table.
int launch_merge(bpe_ctx *c, uint32_t newid, int iter, IterRec *rec, bool with_delta) {
    const uint64_t n = c->n;  // upper bound of the device-side length
    const uint64_t nt = ntiles_of(n);
    TRY(prof_begin(c, BPE_PROF_MERGE, 0));
    if (c->merge_impl == 1) {
        if ((++c->epoch & EPOCH_MASK) == 0) {  // tag wrapped: retire every old descriptor
            HIPCHK(c, hipMemsetAsync(c->d_desc, 0, c->cap_tiles * sizeof(unsigned long long), c->stream));
            HIPCHK(c, hipMemsetAsync(c->d_gdesc, 0, (c->cap_tiles / 64 + 2) * sizeof(unsigned long long), c->stream));
            c->epoch++;
        }
        const unsigned grid = (unsigned)std::max<uint64_t>(nt, 1);
        if (with_delta)
            hipLaunchKernelGGL(k_merge_lookback<true>, dim3(grid), dim3(MT), 0, c->stream,
                               c->d_ids[c->par], c->d_ids[c->par ^ 1], c->d_st, c->par, c->d_desc,
                               c->d_gdesc, c->epoch, newid, c->d_delta, vcap_rep(c), rec, iter, c->d_dirty_n, c->lb_tune);
        else
            hipLaunchKernelGGL(k_merge_lookback<false>, dim3(grid), dim3(MT), 0, c->stream,
                               c->d_ids[c->par], c->d_ids[c->par ^ 1], c->d_st, c->par, c->d_desc,
                               c->d_gdesc, c->epoch, newid, (uint32_t *)nullptr, c->vcap, rec, iter, c->d_dirty_n,
                               c->lb_tune);
        LAUNCHCHK(c, "k_merge_lookback");
    } else {
    if (nt) {
        hipLaunchKernelGGL(k_merge_count, dim3((unsigned)nt), dim3(MT), 0, c->stream,
                           c->d_ids[c->par], c->d_st, c->par, c->d_tsum);
        LAUNCHCHK(c, "k_merge_count");
    }
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, c->stream, c->d_tsum, nt, c->d_tile_off,
                       c->d_tile_sin, c->d_st, c->par, rec, iter, c->d_ids[c->par], c->d_dirty_n);
    LAUNCHCHK(c, "k_tile_scan");
    if (nt) {
        if (with_delta)
            hipLaunchKernelGGL(k_merge_scatter<true>, dim3((unsigned)nt), dim3(MT), 0, c->stream,
                               c->d_ids[c->par], c->d_ids[c->par ^ 1], c->d_st, c->par,
                               c->d_tile_off, c->d_tile_sin, newid, c->d_delta, vcap_rep(c));
        else
            hipLaunchKernelGGL(k_merge_scatter<false>, dim3((unsigned)nt), dim3(MT), 0, c->stream,
                               c->d_ids[c->par], c->d_ids[c->par ^ 1], c->d_st, c->par,
                               c->d_tile_off, c->d_tile_sin, newid, (uint32_t *)nullptr, c->vcap);
        LAUNCHCHK(c, "k_merge_scatter");
    }
    }
    TRY(prof_end(c));
    if (with_delta && c->dp_active) {
        // sharded: fold the replicas into the all-reduce payload; bpe_dp_apply does the rest
        hipLaunchKernelGGL(k_dp_fold, dim3((c->vcap + 255) / 256), dim3(256), 0, c->stream, c->d_delta,
                           vcap_rep(c), newid, c->d_dp_folded);
        LAUNCHCHK(c, "k_dp_fold");
    } else if (with_delta) {
        TRY(prof_begin(c, BPE_PROF_TABLE, 0));
        TRY(launch_table_update<false>(c, c->d_delta, newid, 0, nullptr, 0, 0));
        TRY(prof_end(c));
    }
    c->par ^= 1;
    c->stats_valid = false;
    c->stream_is_bytes = false;
    return BPE_OK;
}


// ---- slotted stream ---------------------------------------------------------------
// contiguous (d_ids[par], st->n[par]) -> slots of TILE ids, all full but the last
int slots_enter(bpe_ctx *c) {
    c->slot_T = ntiles_of(c->n);
    c->mq = 0;
    hipLaunchKernelGGL(k_slot_init, dim3(grid_for(std::max<uint64_t>(c->slot_T, 1), 256, c->num_cus * 4)),
                       dim3(256), 0, c->stream, c->d_meta[0], c->d_hdr[0], c->slot_T, c->d_st, c->par,
                       (uint32_t)c->par, c->d_ids[c->par]);
    LAUNCHCHK(c, "k_slot_init");
    c->slotted = true;
    return BPE_OK;
}

// slots -> contiguous in d_ids[0] (par 0); st->n[0] = the stream length
int slots_leave(bpe_ctx *c) {
    const uint64_t T = c->slot_T;
    if (T) {
        const uint64_t nb = (T + SCAN_TILE - 1) / SCAN_TILE;
        hipLaunchKernelGGL(k_slot_lens, dim3(grid_for(T, 256, c->num_cus * 4)), dim3(256), 0, c->stream,
                           c->d_meta[c->mq], T, c->d_slot_lens);
        hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_slot_lens, T,
                           c->d_slot_bsum);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_slot_bsum, nb,
                           c->d_scratch + 3);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_slot_lens, T,
                           c->d_slot_bsum, c->d_slot_off);
        hipLaunchKernelGGL(k_slot_compact, dim3((unsigned)T), dim3(256), 0, c->stream, c->d_ids[0],
                           c->d_ids[1], c->d_meta[c->mq], c->d_slot_off, c->d_ids2);
        LAUNCHCHK(c, "k_slot_compact");
    }
    std::swap(c->d_ids[0], c->d_ids2);
    if (c->par != 0) {
        hipLaunchKernelGGL(k_move_n, dim3(1), dim3(1), 0, c->stream, c->d_st, c->par, 0);
        LAUNCHCHK(c, "k_move_n");
    }
    c->par = 0;
    c->slotted = false;
    return BPE_OK;
}

// one slotted merge pass + table update (delta mode only)
int launch_merge_slot(bpe_ctx *c, uint32_t newid, int iter, IterRec *rec) {
    TRY(prof_begin(c, BPE_PROF_MERGE, 0));
    if ((++c->epoch & EPOCH_MASK) == 0) {  // tag wrapped: retire every old descriptor
        HIPCHK(c, hipMemsetAsync(c->d_desc, 0, c->cap_tiles * sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_gdesc, 0, (c->cap_tiles / 64 + 2) * sizeof(unsigned long long), c->stream));
        c->epoch++;
    }
    const unsigned slot_grid = (unsigned)std::max<uint64_t>(c->slot_T, 1);
    hipLaunchKernelGGL(k_merge_slot<true>, dim3(slot_grid), dim3(MT), 0,
                       c->stream, c->d_ids[0], c->d_ids[1], c->d_ids[0], c->d_ids[1], c->d_meta[c->mq],
                       c->d_meta[c->mq ^ 1], c->slot_T, c->d_st, c->par, newid, c->d_delta, vcap_rep(c),
                       c->d_dirty_n, c->d_desc, c->epoch, c->d_hdr[c->mq], c->d_hdr[c->mq ^ 1]);
    LAUNCHCHK(c, "k_merge_slot");
    TRY(prof_end(c));
    if (c->dp_active) {
        // sharded: fold the replicas into the all-reduce payload; bpe_dp_apply does the rest
        hipLaunchKernelGGL(k_dp_fold, dim3((c->vcap + 255) / 256), dim3(256), 0, c->stream, c->d_delta,
                           vcap_rep(c), newid, c->d_dp_folded);
        LAUNCHCHK(c, "k_dp_fold");
    } else {
        TRY(prof_begin(c, BPE_PROF_TABLE, 0));
        TRY(launch_table_update<false>(c, c->d_delta, newid, c->par, rec, iter, 1));
        TRY(prof_end(c));
    }
    c->par ^= 1;
    c->mq ^= 1;
    c->stats_valid = false;
    c->stream_is_bytes = false;
    return BPE_OK;
}

int read_state(bpe_ctx *c, DevState *out) {
    HIPCHK(c, hipMemcpyAsync(out, c->d_st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BPE_OK;
}

}  // namespace

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
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    c->own_stream = true;
    for (const void *fn : {(const void *)k_pair_count_lds, (const void *)k_pair_count_bytes})
        if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES)) != hipSuccess)
            return bail("hipFuncSetAttribute(dynamic LDS)", e);
    if ((e = hipMalloc((void **)&c->d_st, sizeof(DevState))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&c->d_scratch, 4 * sizeof(unsigned long long))) != hipSuccess)
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
                    c->d_rowmax, c->d_st,     c->d_tsum,   c->d_tile_off, c->d_tile_sin, c->d_scratch,
                    c->d_delta,  c->d_dirty_list, c->d_dirty_n, c->d_desc, c->d_gdesc, c->d_enc_tmp, c->d_enc_len, c->d_enc_out,
                    c->d_enc_off, c->d_enc_bsum, c->d_enc_long, c->d_ht_keys, c->d_ht_vals, c->d_merge_ids,
                    c->d_dp_folded, c->d_dp_table, c->d_dp_key, c->d_meta[0], c->d_meta[1], c->d_slot_lens,
                    c->d_slot_off, c->d_slot_bsum, c->d_ids2, c->d_hdr[0], c->d_hdr[1],
                    c->d_dec_blob, c->d_dec_out, c->d_dec_voff, c->d_dec_off, c->d_dec_bsum, c->d_dec_ids,
                    c->d_dec_len, c->d_wexp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
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
    } else if (!strcmp(name, "merge")) {
        if (value != 0 && value != 1) return fail(c, BPE_E_ARG, "merge must be 0 or 1");
        c->merge_impl = (int)value;
    } else if (!strcmp(name, "lb_tune")) {
        c->lb_tune = (uint32_t)value;
    } else if (!strcmp(name, "fused_rows")) {
        c->fused_rows = value != 0;
    } else if (!strcmp(name, "slots")) {
        c->use_slots = value != 0;
    } else if (!strcmp(name, "depth")) {
        if (value < 0 || value > 64) return fail(c, BPE_E_ARG, "depth must be 0..64");
        c->depth = (int)value;
    } else {
        return fail(c, BPE_E_ARG, "unknown option '%s'", name);
    }
    return BPE_OK;
}

static int load_bytes_impl(bpe_ctx *c, const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                           uint64_t n_chunks, const uint8_t *wexp) {
    if (!c || (!bytes && n)) return fail(c, BPE_E_ARG, "bytes is NULL");
    if (n >= (1ull << 32)) return fail(c, BPE_E_LIMIT, "stream of %llu bytes exceeds 2^32-1 per GPU",
                                       (unsigned long long)n);
    if (wexp && !chunk_offsets) return fail(c, BPE_E_ARG, "weights need chunk offsets");
    HIPCHK(c, hipSetDevice(c->device));
    if (n + 16 > c->cap_bytes) {
        TRY(dev_realloc(c, c->d_bytes, (size_t)n + 16));
        c->cap_bytes = n + 16;
    }
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_bytes, bytes, n, hipMemcpyHostToDevice, c->stream));
    static const uint64_t zero = 0;
    if (!chunk_offsets) {
        chunk_offsets = &zero;
        n_chunks = 1;
    }
    if (n_chunks > c->cap_offsets) {
        TRY(dev_realloc(c, c->d_offsets, (size_t)n_chunks));
        c->cap_offsets = n_chunks;
    }
    if (n_chunks)
        HIPCHK(c, hipMemcpyAsync(c->d_offsets, chunk_offsets, n_chunks * sizeof(uint64_t),
                                 hipMemcpyHostToDevice, c->stream));
    c->weighted = false;
    if (wexp && n_chunks) {
        for (uint64_t i = 0; i < n_chunks; i++)
            if (wexp[i] > 31) return fail(c, BPE_E_ARG, "weight exponent %u of chunk %llu exceeds 31", wexp[i],
                                          (unsigned long long)i);
        if (n_chunks > c->cap_wexp) {
            TRY(dev_realloc(c, c->d_wexp, (size_t)n_chunks));
            c->cap_wexp = n_chunks;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_wexp, wexp, n_chunks, hipMemcpyHostToDevice, c->stream));
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

int bpe_train(bpe_ctx *c, int32_t num_merges, int32_t *pairs_out, uint64_t *counts_out,
              double *iter_ms_out, uint64_t *len_out, int32_t *n_done) {
    if (!c || num_merges < 0) return fail(c, BPE_E_ARG, "bad arguments");
    if (!c->have_bytes) return fail(c, BPE_E_STATE, "bpe_load_bytes first");
    if (c->dp_active) return fail(c, BPE_E_STATE, "bpe_dp_end first");
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
    TRY(prof_end(c));
    if (iter_ms_out) HIPCHK(c, hipEventRecord(evs[0], c->stream));
    const uint64_t n0 = c->n;
    TRY(launch_pair_count(c, false));

    int done = 0, rc = BPE_OK, consumed = 0;
    uint64_t cur_len = n0;  // exact length before iteration `consumed`
    bool stop = false;
    c->rep_shift = 5;
    const bool slots = delta && c->use_slots && c->merge_impl == 0;
    if (slots) TRY(slots_enter(c));
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
        // sites per pass ~ the merged pair's count: fewer sites, fewer replicas to fold
        // few sites -> few same-address atomics -> fewer replicas to fold (measured: going
        // below 32 while a pass still has tens of thousands of sites slows the merge pass)
        c->rep_shift = 5;  // (k_apply_delta folds 32 replicas with 16 loads in flight per lane: no need to shrink)
        done = j + 1;
        return BPE_OK;
    };

    int i = 0;
    while (!stop) {
        // enqueue iteration i (if any is left), then look at the record `depth` back
        if (i < num_merges) {
            c->vcur = 256u + (uint32_t)i;
            bool full_rowmax = (i == 0);
            if (!delta && i > 0) {
                TRY(clear_table(c));
                TRY(launch_pair_count(c, false));
                full_rowmax = true;
            }
            // Slots thinning out: re-pack (between merges nothing is pending).  A pass costs
            // per slot as much as per id, so the slot count should follow the stream length
            // closely; at 31/32 fill a whole cfg2 run re-packs ~45 times, ~60 us each.
            if (c->slotted && c->slot_T > 64 &&
                c->n * REPACK_DEN < c->slot_T * (uint64_t)TILE * (REPACK_DEN - 1)) {
                TRY(slots_leave(c));
                TRY(slots_enter(c));
            }
            TRY(launch_select(c, full_rowmax));
            if (c->slotted)
                TRY(launch_merge_slot(c, 256u + (uint32_t)i, i, c->h_rec));
            else
                TRY(launch_merge(c, 256u + (uint32_t)i, i, c->h_rec, delta));
            if (iter_ms_out) HIPCHK(c, hipEventRecord(evs[(size_t)i + 1], c->stream));
            i++;
        }
        if (consumed < i && (i - consumed > c->depth || i == num_merges)) {
            TRY(consume(consumed));
            if (!stop) consumed++;
        }
        if (consumed >= num_merges) break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->slotted) {
        // leave the ids contiguous for whoever reads them next
        if (stop) {  // parity of the no-op iterations enqueued after the failing one
            const int back = i - done;
            if (back & 1) {
                c->par ^= 1;
                c->mq ^= 1;
            }
            hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, c->stream, c->d_st, 0u);
        }
        TRY(slots_leave(c));
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
    if (n_done) *n_done = done;
    return rc;
}

}  // extern "C" (reopened below)

// ---------------------------------------------------------------------------
// encode

namespace {
inline uint64_t mix_key(uint64_t key) { return (key * 0x9E3779B97F4A7C15ull) >> 40; }

int upload_offsets(bpe_ctx *c, const uint64_t *chunk_offsets, uint64_t n_chunks) {
    if (n_chunks > c->cap_offsets) {
        TRY(dev_realloc(c, c->d_offsets, (size_t)n_chunks));
        c->cap_offsets = n_chunks;
    }
    if (n_chunks)
        HIPCHK(c, hipMemcpyAsync(c->d_offsets, chunk_offsets, n_chunks * sizeof(uint64_t),
                                 hipMemcpyHostToDevice, c->stream));
    return BPE_OK;
}
}  // namespace

extern "C" int bpe_encode_batch(bpe_ctx *c, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                                const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                                uint64_t n_chunks, int32_t *ids_out, uint64_t *out_offsets,
                                uint64_t *n_out) {
    if (!c || M < 0 || (!merges && M) || (!bytes && n)) return fail(c, BPE_E_ARG, "bad arguments");
    if (n >= (1ull << 32)) return fail(c, BPE_E_LIMIT, "batch of %llu bytes exceeds 2^32-1", (unsigned long long)n);
    static const uint64_t zero = 0;
    if (!chunk_offsets) {
        chunk_offsets = &zero;
        n_chunks = 1;
    }
    if (n_out) *n_out = 0;
    if (n == 0 || n_chunks == 0) {
        if (out_offsets)
            for (uint64_t i = 0; i <= n_chunks; i++) out_offsets[i] = 0;
        return BPE_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipMemcpyAsync(c->d_bytes, bytes, n, hipMemcpyHostToDevice, c->stream));
    TRY(upload_offsets(c, chunk_offsets, n_chunks));
    // 3. scratch
    if (n > c->cap_enc_n) {
        TRY(dev_realloc(c, c->d_enc_tmp, (size_t)n));
        TRY(dev_realloc(c, c->d_enc_out, (size_t)n));
        TRY(dev_realloc(c, c->d_enc_long, (size_t)(n / (ENC_LMAX + 1) + 2)));
        c->cap_enc_n = n;
    }
    const uint64_t nb = (n_chunks + SCAN_TILE - 1) / SCAN_TILE;
    if (n_chunks > c->cap_enc_chunks) {
        TRY(dev_realloc(c, c->d_enc_len, (size_t)n_chunks));
        TRY(dev_realloc(c, c->d_enc_off, (size_t)n_chunks + 1));
        TRY(dev_realloc(c, c->d_enc_bsum, (size_t)nb + 1));
        c->cap_enc_chunks = n_chunks;
    }
    unsigned long long *d_nlong = c->d_scratch, *d_total = c->d_scratch + 1;
    uint32_t *d_min = (uint32_t *)(c->d_scratch + 2);
    HIPCHK(c, hipMemsetAsync(c->d_scratch, 0, 2 * sizeof(unsigned long long), c->stream));
    const uint32_t mask = (uint32_t)(hs - 1);
    // 4. one chunk per lane
    TRY(prof_begin(c, BPE_PROF_ENCODE, n));
    hipLaunchKernelGGL(k_encode_short, dim3((unsigned)((n_chunks + ENC_THREADS - 1) / ENC_THREADS)),
                       dim3(ENC_THREADS), 0, c->stream, c->d_bytes, c->d_offsets, n_chunks, n,
                       c->d_ht_keys, c->d_ht_vals, mask, d_mids, c->d_enc_tmp, c->d_enc_len,
                       c->d_enc_long, d_nlong);
    LAUNCHCHK(c, "k_encode_short");
    TRY(prof_end(c));
    unsigned long long n_long = 0;
    HIPCHK(c, hipMemcpyAsync(&n_long, d_nlong, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // 5. the few long chunks: stream-wide rounds (lowest rank present -> merge everywhere)
    if (n_long) {
        std::vector<unsigned long long> ids_l(n_long);
        HIPCHK(c, hipMemcpy(ids_l.data(), c->d_enc_long, n_long * 8, hipMemcpyDeviceToHost));
        std::sort(ids_l.begin(), ids_l.end());
        std::vector<unsigned long long> src(n_long), dst(n_long + 1);
        unsigned long long tot = 0;
        for (uint64_t k = 0; k < n_long; k++) {
            const uint64_t ch = ids_l[k];
            const uint64_t s0 = chunk_offsets[ch], e0 = (ch + 1 < n_chunks) ? chunk_offsets[ch + 1] : n;
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
    // 6. output offsets = exclusive scan of the per-chunk lengths
    hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_len,
                       n_chunks, c->d_enc_bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_enc_bsum, nb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_enc_len, n_chunks,
                       c->d_enc_bsum, c->d_enc_off);
    LAUNCHCHK(c, "k_scan_*");
    // 7. placement and copy-out
    hipLaunchKernelGGL(k_encode_place, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_enc_tmp, c->d_offsets, c->d_enc_len, c->d_enc_off, n_chunks, c->d_enc_out);
    LAUNCHCHK(c, "k_encode_place");
    unsigned long long total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ids_out && total)
        HIPCHK(c, hipMemcpy(ids_out, c->d_enc_out, total * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_offsets) {
        HIPCHK(c, hipMemcpy(out_offsets, c->d_enc_off, n_chunks * 8, hipMemcpyDeviceToHost));
        out_offsets[n_chunks] = total;
    }
    if (n_out) *n_out = total;
    TRY(prof_drain(c));
    return BPE_OK;
}

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
    TRY(ensure_rec(c, std::max(num_merges, 1)));
    memset(c->h_rec, 0, sizeof(IterRec) * (size_t)std::max(num_merges, 1));
    if (c->d_dp_folded) (void)hipFree(c->d_dp_folded);
    c->d_dp_folded = nullptr;
    HIPCHK(c, hipMalloc((void **)&c->d_dp_folded, (size_t)c->vcap * 4 * sizeof(uint32_t)));
    if (!c->d_dp_table) HIPCHK(c, hipMalloc((void **)&c->d_dp_table, 256 * 256 * sizeof(uint32_t)));
    if (!c->d_dp_key) HIPCHK(c, hipMalloc((void **)&c->d_dp_key, 2 * sizeof(long long)));
    TRY(start_from_bytes(c));
    HIPCHK(c, hipMemsetAsync(c->d_mat, 0, (size_t)c->vcap * c->vcap * sizeof(uint32_t), c->stream));
    TRY(launch_pair_count(c, false));
    // the byte-pair block of the table, packed, is the first all-reduce payload
    HIPCHK(c, hipMemcpy2DAsync(c->d_dp_table, 256 * 4, c->d_mat, (size_t)c->vcap * 4, 256 * 4, 256,
                               hipMemcpyDeviceToDevice, c->stream));
    c->dp_cur_len = c->n;
    c->dp_enq = c->dp_done = 0;
    c->rep_shift = 5;
    if (c->use_slots) TRY(slots_enter(c));
    return BPE_OK;
}

extern "C" int bpe_dp_buffers(bpe_ctx *c, void **table, uint64_t *table_count, void **delta,
                              uint64_t *delta_count, void **tiekey) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    if (table) *table = c->d_dp_table;
    if (table_count) *table_count = 256 * 256;
    if (delta) *delta = c->d_dp_folded;
    if (delta_count) *delta_count = (uint64_t)c->vcap * 4;
    if (tiekey) *tiekey = c->d_dp_key;
    return BPE_OK;
}

extern "C" int bpe_dp_table_ready(bpe_ctx *c) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(c->d_mat, (size_t)c->vcap * 4, c->d_dp_table, 256 * 4, 256 * 4, 256,
                               hipMemcpyDeviceToDevice, c->stream));
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
    if (c->slotted && c->slot_T > 64 &&
        c->n * REPACK_DEN < c->slot_T * (uint64_t)TILE * (REPACK_DEN - 1)) {
        TRY(slots_leave(c));
        TRY(slots_enter(c));
    }
    TRY(launch_select(c, false));
    hipLaunchKernelGGL(k_dp_key, dim3(1), dim3(64), 0, c->stream, stream_ref(c), c->par, c->d_st,
                       (unsigned long long)c->dp_rank, c->d_dp_key);
    LAUNCHCHK(c, "k_dp_key");
    return BPE_OK;
}

extern "C" int bpe_dp_merge(bpe_ctx *c, int32_t iter) {
    if (!c || !c->d_dp_folded) return fail(c, BPE_E_STATE, "bpe_dp_begin first");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_dp_resolve, dim3(1), dim3(64), 0, c->stream, c->d_st, c->d_dp_key);
    LAUNCHCHK(c, "k_dp_resolve");
    c->dp_enq = iter + 1;
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
        if ((c->dp_enq - c->dp_done) & 1) {
            c->par ^= 1;
            c->mq ^= 1;
        }
        hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, c->stream, c->d_st, 0u);
        TRY(slots_leave(c));
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
    for (int i = 0; i < num_merges && !stop; i++) {
        TRY(bpe_dp_select(c, i));
        RCCLCHK(c, r->AllReduce(c->d_dp_key, c->d_dp_key, 2, RCCL_INT64, RCCL_MIN, c->comm, c->stream));
        TRY(bpe_dp_merge(c, i));
        RCCLCHK(c, r->AllReduce(c->d_dp_folded, c->d_dp_folded, (size_t)c->vcap * 4, RCCL_INT32, RCCL_SUM,
                                c->comm, c->stream));
        TRY(bpe_dp_apply(c, i));
        // the schedule depends on i only: every rank issues the same collectives even when one stops
        if (i - consumed >= c->depth) {
            TRY(consume(consumed));
            if (!stop) consumed++;
        }
    }
    while (!stop && consumed < num_merges) {
        TRY(consume(consumed));
        if (!stop) consumed++;
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

extern "C" {

int bpe_prof_reset(bpe_ctx *c) {
    if (!c) return BPE_E_ARG;
    TRY(prof_drain(c));
    for (int k = 0; k < BPE_PROF_NKINDS; k++) {
        c->prof_ms[k] = 0;
        c->prof_launches[k] = 0;
        c->prof_bytes[k] = 0;
    }
    return BPE_OK;
}

int bpe_prof_read(bpe_ctx *c, double *ms, uint64_t *launches, uint64_t *alg_bytes) {
    if (!c) return BPE_E_ARG;
    TRY(prof_drain(c));
    for (int k = 0; k < BPE_PROF_NKINDS; k++) {
        if (ms) ms[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_launches[k];
        if (alg_bytes) alg_bytes[k] = c->prof_bytes[k];
    }
    return BPE_OK;
}

}  // extern "C"
