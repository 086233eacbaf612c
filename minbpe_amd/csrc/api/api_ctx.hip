// api_ctx.hip -- the ctx, error / allocation helpers, profiling events, kernel-launch helpers.
// Part of bpe_api.hip, which includes the parts in order (one translation unit).

// how a sharded training loop reaches the other ranks (bpe_dp_train: librccl on the ctx's stream; bpe_dp_train_cb:
// the caller's function)
struct DpComm {
    bpe_allreduce_fn fn = nullptr;
    void *user = nullptr;
    int rank = 0, nranks = 1;
};

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
    // sharded chain steps (dp_train_loop): the communicator of the loop in progress, the MIN payload of a tie's first
    // occurrences (DP_KEY_WORDS int64), the SUM payload of a batch's delta (2 dp_kcap vcap + 64 words)
    const struct DpComm *dp_comm = nullptr;
    long long *d_dp_ckey = nullptr;
    uint32_t *d_dp_cfold = nullptr;
    uint64_t cap_dp_cfold = 0;
    bool dp_force_comm = false;  // option "dp_force_comm": issue the collectives in a world of one too (tests, launch-cost measurements)
    int dp_kcap = DP_KCAP_DEFAULT;  // option "dp_kcap": most pairs of a sharded step's batch (the SUM payload, 2 dp_kcap S words, grows with it)
    uint64_t *d_round_lb = nullptr;  // k_load_count: first chunk of every segment of the byte stream
    uint64_t cap_round_lb = 0;
    bool fuse_load = true;  // option "fuse_load": the byte stream's first get_stats rides on the widening pass
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
    unsigned long long *d_sup = nullptr;  // streams of many tiles: the transducers of the super-tiles and their prefixes (k_tile_sup)
    uint64_t cap_tiles = 0;
    IterRec *h_rec = nullptr;  // pinned, device-visible
    int rec_cap = 0;
    unsigned long long *d_scratch = nullptr;  // 2 x u64 cursor/counter
    uint32_t *d_delta = nullptr;       // 4 x vcap: decL | decR | incL | incR
    uint32_t *d_dirty_list = nullptr;  // rows whose rowmax must be recomputed
    uint32_t *d_dirty_n = nullptr;
    int depth = 8;  // iterations the host may run ahead of the device
    int repack_acc = 200;  // option "repack_acc": 0 = re-pack the dense phase's slots when their fill drops below 31/32; N = when the
                         // empty fractions of the sweeps since the last re-pack add up to N/100 of a sweep (what a re-pack costs)
    double repack_waste = 0.0;
    // slotted stream (training loop, a != b merges)
    int use_slots = 2;  // 0 contiguous | 1 slotted, first form (k_slots.hip) | 2 second form (k_slots2.hip)
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
    // second slotted form (k_slots2.hip; use_slots == 2, the default)
    bool slot2 = false;                       // the slotted stream currently uses the 32-byte headers
    SlotHdr *d_hdr2[2] = {nullptr, nullptr};
    StageRec *d_stage = nullptr;              // staged headers of a sparse pass: stage[t] for slot t
    uint32_t *d_smask = nullptr;              // [slot / 32] which slots have a staged header
    uint32_t *d_cand = nullptr;               // candidate slots of a sparse pass (made by k_select)
    uint32_t *d_idx = nullptr;                // inverted slot index [IDX_H][idx_cap_words] (bucket-major)
    uint32_t *d_idx_tmp = nullptr;            // group-major image the build kernel writes, transposed into d_idx
    uint32_t *d_idx_dirty = nullptr;          // [slot / 32]: slots rewritten by a == b passes since the last build
    uint32_t *d_removed = nullptr;            // [256 * REMOVED_STRIDE] removal counters of a merge pass
    uint64_t idx_cap_words = 0;               // index groups allocated
    uint64_t idx_cap_rows = 0;                // ... and buckets (rows): those of the geometry the index was allocated in
    bool idx_rebuild = false;                 // an a == b merge went by: rebuild before the next pass
    bool last_aa_indexed = false;             // the last general-path iteration enqueued had its a == b pass keep the index current
    bool idx_live = false;                    // the index describes the current slots
    int use_sparse = 1;                       // 0: never take the sparse pass (experiments)
    int rep_max = 8;                          // log2 of the most delta-vector replicas a pass may use (experiments)
    int rep_min = 4;                          // option "rep_min": log2 of the fewest delta replicas a pass uses (a hot token's
                                              // atomics queue ~11 ns apiece per replica: 16 replicas beat 1 by ~2 us per late pass)
    int lds_delta = 1;                        // option "lds_delta": a != b passes aggregate their delta in LDS while ids < LDSD_CAP
    int exp_no_delta = 0;                     // experiment: a != b passes skip the pair-table bookkeeping (wrong results)
    int tie_window = 0;                       // block 0 sweeps the first slots alone on a tie (measured slower: off)
    int tie_index = 1;                        // break ties through the index when it is live (0: always sweep)
    int sparse_ratio = 1;                     // sparse pass when (count of the pair) * ratio < slots (measured at 1 GB: 1 beats 2 by 30 us per merge over iterations 300-1000)
    uint64_t last_count = ~0ull;              // count of the last merge the host has seen: an upper bound of the next ones
    uint64_t n_sparse = 0, n_dense = 0, n_index_builds = 0;  // passes of the last train() (bpe_train_stats)
    uint64_t n_lean = 0, n_deferred = 0;      // ... of which lean iterations (k_lean.hip); iterations handed back to the general path
    int lean = 1;                             // option "lean": 0 never | 1 once the last seen count is <= lean_count | 2 always (tests)
    int64_t lean_count = 1 << 20;             // option "lean_count"
    int lean_backoff = 1;                     // option "lean_backoff": general-path stretches after clustered deferrals (api_train.hip)
    int lean_grid = 256;                      // option "lean_grid": most workgroups of a lean merge pass
    int lean_scan = 31;                       // option "lean_scan": workgroups of k_rowmax_lean
    int lean_select = 1;                      // option "lean_select": 1 = k_rowsel_lean (row maxima + selection in one launch) while the index is live
    int aa_sparse = 1;                        // option "aa_sparse": a sparse iteration's a == b pass works through a candidate list and keeps the index current itself (no rebuild after it)
    int lean_chain = 1;                       // option "lean_chain": 1 = tied pairs are merged off the list one selection made (k_sel_lean), 0 = every iteration selects
    int chain_dense = 1;                      // option "chain_dense": chain steps from the second merge on -- dense passes, LDS delta tables, up to CH_KDENSE
                                              // pairs per sweep -- while every id is below LDSD_CAP and the index does not exist yet
    unsigned long long *d_chain_req = nullptr;  // the request / answer words between the deciding and the scanning workgroups of a chain step's selection (k_pool.hip)
    int ts = TILE2_MAX;                       // ids per slot of the second slotted form as the stream stands: 1024 (kernels of namespace bpe_g4)
                                              // or 256 (bpe_g1) -- see GK below
    int small_slots = 1;                      // option "small_slots": re-pack into 256-id slots when the inverted index is first built (a large
                                              // stream going sparse: a merge site then costs 1 KiB of its slot, not 4); 2 = streams of a few thousand
                                              // slots too (tests)
    int count_is_removed = 1;                 // option "count_is_removed": chain steps of an unweighted stream take the ids a merge removes from the pair's
                                              // count instead of counting them (0: count, the cross-check)
    int chain_prefetch = 1;                   // option "chain_prefetch": 256-id slots -- a wave's next candidate slot is loaded while it works on this one
    int chain_kcap = CH_KSWEEP;               // option "chain_kcap": most pairs of a sparse chain step's batch (1..CH_KSWEEP)
    int fuse_step = 0;                        // option "fuse_step": 1 = a sparse chain step of a single-GPU job is ONE launch (k_step.hip:
                                              // selection -> published batch -> merge pass -> grid barrier -> table update) instead of three.
                                              // Measured SLOWER than the three launches (this part overlaps a dependent launch's
                                              // ramp with its predecessor's tail -- median gap 0.0 us -- and a fence-free grid barrier is 3.8 us): off
    unsigned long long *d_step_pub = nullptr; // ... its published line (STEP_PUB_WORDS granules)
    uint32_t *d_step_bar = nullptr;           // ... its grid-barrier counter (only ever grows; zeroed when a train() begins)
    uint32_t step_bar_target = 0;             // ... and what it will read once every launch enqueued so far is through its barrier
    // encode as a replay of training (api_encode.hip): the loop's selections are the merges of d_forced, in order
    bool forced = false;
    int32_t *d_forced = nullptr;
    uint64_t cap_forced = 0;
    int enc_replay = 1;                       // option "enc_replay": one giant chunk (BasicTokenizer.encode) is encoded by replaying its merge
                                              // list through the training engine (0: the stream-wide rounds)
    // host -> device uploads of large caller buffers (upload_h2d below): a ring of pinned staging buffers filled by several
    // host threads while the copy engine drains the ones before (option "pinned_upload": 0 = plain hipMemcpyAsync)
    int pinned_upload = 1;
    uint8_t *h_stage = nullptr;
    hipEvent_t ev_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_busy[4] = {false, false, false, false};  // a copy out of / into the buffer was enqueued: wait for its event before the buffer is touched again
    uint64_t n_fused = 0;                     // chain steps of the last train() that were one launch
    unsigned long long *d_step_stamps = nullptr;  // debug (env BPE_STEP_STAMPS=file): clock stamps of its phases, dumped when train() ends
    int pool_hint = 0;                        // option "pool_hint": a rebuild is announced when fewer untouched entries than this are left (0: the step's cap)
    PoolEnt *d_pool = nullptr;                // ... its entries (PL_CAP) and the pairs a rebuild gathers (counter, pad, PL_GATHER x {pair, count})
    uint32_t *d_pool_gather = nullptr;
    int chain_scan = 127;                     // option "chain_scan": workgroups that re-scan flagged rows in a chain step's FULL selection (a level of
                                              // n merges leaves ~3 n rows to re-scan, one 128 KB row per workgroup at a time)
    int chain = 1;                            // option "chain": 1 = chain steps (k_chain.hip: the tied pairs kept as a list, batches of
                                              // token-disjoint pairs merged in one pass) instead of lean iterations, wherever those would run with the index live
    StepRec *h_srec = nullptr;                // pinned ring of step records (STEP_RING entries)
    uint64_t n_steps = 0, n_full = 0, n_chained = 0;  // chain steps of the last train(), those that selected, merges that needed no selection
    int lean_sum = 1;                         // option "lean_sum": 1 = k_sel_lean (selection from the table update's per-wave records) whenever they are current
    uint4 *d_lean_sum = nullptr;              // [4][LEAN_SUM_CAP] those records (k_lean.hip)
    bool sum_valid = false;                   // ... describe the table as it stands (the last iteration enqueued was a lean one)
    uint32_t *d_dbits = nullptr;              // [DBITS_WORDS] rows a lean table update flagged for re-scanning
    unsigned long long *d_lean_res = nullptr; // row maxima on their way to the deciding workgroup of k_rowsel_lean: 2 words per item
    uint32_t lean_tag = 0;                    // ... tagged with this launch counter
    bool rows_pending = false;                // a lean table update ran and its rows have not been re-scanned yet
    uint64_t cap_slots = 0;
    // data-parallel stepping (bpe_dp_*)
    int dp_rank = 0, dp_nranks = 1, dp_merges = 0, dp_enq = 0, dp_done = 0;
    bool dp_active = false;  // between bpe_dp_begin and bpe_dp_end
    void *comm = nullptr;    // RCCL communicator (bpe_comm_init), one rank per ctx
    int comm_rank = 0, comm_nranks = 1;
    uint32_t *d_dp_folded = nullptr, *d_dp_table = nullptr;
    long long *d_dp_key = nullptr;
    uint64_t dp_cur_len = 0;
    bool dp_sparse = false;                   // second slotted form: this iteration's a != b pass is a sparse one
    uint32_t dp_dl = 0;                       // ... and the delta layout its kernels used
    std::vector<uint8_t> dp_flip;             // ... and which iterations flipped the header arrays
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
    unsigned long long *d_enc_huge = nullptr;  // chunks of more than ENC_LONG_TOP bytes: the stream-wide rounds' (k_enc_long hands them on)
    int enc_long = 1;                          // option "enc_long": chunks of ENC_LMAX + 1 .. ENC_LONG_TOP bytes are encoded on the device, one wave
                                               // per chunk (k_enc_long); 0 = all of them through the stream-wide rounds (round 4's path, a cross-check)
    unsigned long long *d_ht_keys = nullptr;
    uint32_t *d_ht_vals = nullptr;
    int32_t *d_merge_ids = nullptr;
    uint64_t cap_enc_n = 0, cap_enc_chunks = 0, cap_ht = 0, cap_merge_ids = 0;
    // the chunk cache of bpe_encode_batch (k_encode.hip): hash table + per chunk its slot, then its owner
    EncEntry *d_enc_tab = nullptr;
    uint32_t *d_enc_rep = nullptr;  // per chunk: its slot
    uint8_t *d_enc_mid = nullptr;             // pass 2's work list: lane numbers, packed per workgroup of pass 1 (k_encode.hip)
    uint32_t *d_enc_midn = nullptr;           // ... and how many per workgroup
    uint64_t cap_enc_tab = 0;
    int enc_cache = 1;  // option "enc_cache": 0 = encode every chunk on its own
    int enc_chain = 1;  // option "enc_chain": 1 = output offsets and placement in one chained pass (k_enc_place_chained)
    int enc_hash_bits = 0;  // option "enc_hash_bits" (tests): keep only this many bits of the chunk hash (0 = all 64)

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
    // An event record drains the queue: a timed iteration costs ~30 us more than an untimed one (measured:
    // +9 % on a whole 1 GB train with every 8th late iteration timed).  Inside bpe_train the first
    // PROF_FULL_ITERS iterations -- the long ones, where an event is free -- are all timed, later ones every
    // prof_stride-th, weighted by the stride.
    int prof_iter = -1;   // iteration being enqueued by bpe_train (-1: not in its loop)
    int prof_stride = 64;  // option "prof_stride"
    bool prof_active = false;
    int scan_sup_min = 1024;  // the three-pass merge's tile scan: more tiles than this take k_tile_sup / k_tile_scan_sup / k_tile_expand (option scan_sup: 0 = always, 1 << 30 = never)
    int k1 = 2;       // 0 one atomic per position | 1 LDS hash cache (8-byte slots) | 2 = 1 + dense 16-bit LDS table for byte streams | 3 = 2 with the 4-byte-slot LDS cache for general unweighted streams (measured: no faster, both bound by L2 atomics on cold pairs)
    bool stream_is_bytes = false;  // every id of the current stream is < 256 (fresh from k_widen)

    std::vector<ProfEv> prof_open;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[BPE_PROF_NKINDS] = {0};
    uint64_t prof_launches[BPE_PROF_NKINDS] = {0};
    uint64_t prof_bytes[BPE_PROF_NKINDS] = {0};
};

// the kernel of the geometry the slotted stream is in (bpe_device.h: BPE_GEOMETRY; kernels of one name have one signature)
#define GK(c, kern) ((c)->ts == TILE2_MIN ? bpe::bpe_g1::kern : bpe::bpe_g4::kern)
inline uint32_t idx_h(const bpe_ctx *c) { return c->ts == TILE2_MIN ? bpe::bpe_g1::IDX_H : bpe::bpe_g4::IDX_H; }

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
    const uint64_t nt2 = (TILE / TILE2_MIN) * nt + 4;  // slots of the second slotted form (as few as TILE2_MIN ids each)
    if (nt > c->cap_tiles) {
        TRY(dev_realloc(c, c->d_tsum, nt));
        TRY(dev_realloc(c, c->d_tile_off, nt));
        TRY(dev_realloc(c, c->d_tile_sin, nt));
        TRY(dev_realloc(c, c->d_sup, 2 * 3 * (nt / SUP_TILES + 2)));  // (a TS is three 8-byte words)
        TRY(dev_realloc(c, c->d_meta[0], nt));
        TRY(dev_realloc(c, c->d_meta[1], nt));
        TRY(dev_realloc(c, c->d_hdr[0], nt));
        TRY(dev_realloc(c, c->d_hdr[1], nt));
        TRY(dev_realloc(c, c->d_hdr2[0], nt2));
        TRY(dev_realloc(c, c->d_hdr2[1], nt2));
        TRY(dev_realloc(c, c->d_stage, nt2));
        TRY(dev_realloc(c, c->d_smask, nt2 / 32 + 2));
        TRY(dev_realloc(c, c->d_cand, nt2));
        HIPCHK(c, hipMemsetAsync(c->d_smask, 0, (nt2 / 32 + 2) * sizeof(uint32_t), c->stream));
        TRY(dev_realloc(c, c->d_slot_lens, nt2));
        TRY(dev_realloc(c, c->d_slot_off, nt2 + 1));
        TRY(dev_realloc(c, c->d_slot_bsum, nt2 / SCAN_TILE + 2));
        TRY(dev_realloc(c, c->d_desc, nt2));
        TRY(dev_realloc(c, c->d_gdesc, nt / 64 + 2));
        HIPCHK(c, hipMemsetAsync(c->d_desc, 0, nt2 * sizeof(unsigned long long), c->stream));
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
    TRY(dev_realloc(c, c->d_rowmax, (size_t)nv * 2));  // interleaved: rowmax[2x] = max of row x, rowmax[2x + 1] = its column (or ROWARG_MULTI)
    TRY(dev_realloc(c, c->d_delta, (size_t)nv * 4 * DELTA_REPL + 256 * DELTA_SKEW));  // (+ the skew of up to 256 replicas)
    TRY(dev_realloc(c, c->d_dirty_list, (size_t)nv));
    if (!c->d_dirty_n) HIPCHK(c, hipMalloc((void **)&c->d_dirty_n, sizeof(uint32_t)));
    if (!c->d_dbits) {
        HIPCHK(c, hipMalloc((void **)&c->d_dbits, DBITS_WORDS * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(c->d_dbits, 0, DBITS_WORDS * sizeof(uint32_t), c->stream));
    }
    TRY(dev_realloc(c, c->d_lean_res, 2 * ((size_t)nv + 8)));
    if (!c->d_lean_sum) HIPCHK(c, hipMalloc((void **)&c->d_lean_sum, 4 * (size_t)LEAN_SUM_CAP * sizeof(uint4)));
    HIPCHK(c, hipMemsetAsync(c->d_lean_res, 0, 2 * ((size_t)nv + 8) * sizeof(unsigned long long), c->stream));
    if (!c->d_removed) {
        HIPCHK(c, hipMalloc((void **)&c->d_removed, 256 * REMOVED_STRIDE * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(c->d_removed, 0, 256 * REMOVED_STRIDE * sizeof(uint32_t), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->d_delta, 0, ((size_t)nv * 4 * DELTA_REPL + 256 * DELTA_SKEW) * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_dirty_n, 0, sizeof(uint32_t), c->stream));
    // rows beyond the ids in use read as "no pair" (a deferred lean iteration's no-op successors run
    // k_select with a vocabulary the table has not reached yet)
    HIPCHK(c, hipMemsetAsync(c->d_rowmax, 0, (size_t)nv * 2 * sizeof(uint32_t), c->stream));
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

int ensure_srec(bpe_ctx *c) {
    if (!c->h_srec) HIPCHK(c, hipHostMalloc((void **)&c->h_srec, sizeof(StepRec) * STEP_RING, hipHostMallocMapped));
    memset(c->h_srec, 0, sizeof(StepRec) * STEP_RING);
    return BPE_OK;
}

// ---- profiling --------------------------------------------------------------
constexpr int PROF_FULL_ITERS = 1024;
int prof_begin(bpe_ctx *c, int kind, uint64_t bytes) {
    // level 1: only the dominant kernel class (merge) -- two event records per
    // iteration; level 2: every class (adds marker packets between all kernels)
    c->prof_active = c->profile >= 2 || (c->profile == 1 && kind == BPE_PROF_MERGE);
    if (!c->prof_active) return BPE_OK;
    uint32_t weight = 1;
    if (c->prof_iter >= PROF_FULL_ITERS && c->prof_stride > 1) {
        if (c->prof_iter % c->prof_stride) {
            c->prof_active = false;
            return BPE_OK;
        }
        weight = (uint32_t)c->prof_stride;
    }
    ProfEv ev;
    ev.kind = kind;
    ev.bytes = bytes;
    ev.weight = weight;
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
        c->prof_ms[ev.kind] += (double)ms * ev.weight;
        c->prof_launches[ev.kind] += ev.weight;
        c->prof_bytes[ev.kind] += ev.bytes * ev.weight;
        c->ev_pool.push_back(ev.e0);
        c->ev_pool.push_back(ev.e1);
    }
    c->prof_open.clear();
    return BPE_OK;
}

// one collective of the sharded loop in progress, enqueued behind the work already on the ctx's stream
int dp_allreduce(bpe_ctx *c, void *buf, uint64_t count, int32_t dtype, int32_t op) {
    const DpComm *dp = c->dp_comm;
    if (!dp || !dp->fn) return fail(c, BPE_E_STATE, "no communicator");
    if (dp->nranks == 1 && !c->dp_force_comm) return BPE_OK;  // (a world of one: the buffer is its own reduction)
    const int rc = dp->fn(dp->user, buf, count, dtype, op, (void *)c->stream);
    if (rc != 0) return fail(c, BPE_E_HIP, "all-reduce failed (%d)", rc);
    return BPE_OK;
}

// ---- launch helpers -----------------------------------------------------------
inline unsigned grid_for(uint64_t work_items, unsigned per_block, unsigned cap) {
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// The byte stream's first get_stats can ride on the widening pass (k_load_count): unit increments into 16-bit LDS
// counters, so not for weighted chunks; option "fuse_load" = 0 keeps the three separate passes (cross-check).
inline bool load_count_fusable(const bpe_ctx *c) { return c->fuse_load && c->k1 == 2 && !c->weighted && c->nbytes >= 2; }

// widen resident bytes into ids[0], mark chunk starts, reset state.  count = true (the table is cleared, on this
// stream, and load_count_fusable(c) holds): the pair counts of the byte stream go into the table in the same pass.
int start_from_bytes(bpe_ctx *c, bool count = false) {
    const uint64_t n = c->nbytes;
    TRY(ensure_ids(c, n));
    if (count) {
        const uint64_t segs = (n + LC_SEG - 1) / LC_SEG;
        if (segs + 2 > c->cap_round_lb) {
            TRY(dev_realloc(c, c->d_round_lb, (size_t)segs + 2));
            c->cap_round_lb = segs + 2;
        }
        TRY(prof_begin(c, BPE_PROF_PAIR_COUNT, 4 * n));
        const uint64_t *off = c->n_chunks ? c->d_offsets : nullptr;
        if (off) {
            hipLaunchKernelGGL(k_seg_lb, dim3((unsigned)((segs + 256) / 256)), dim3(256), 0, c->stream, off,
                               (uint64_t)c->n_chunks, segs, c->d_round_lb);
            LAUNCHCHK(c, "k_seg_lb");
        }
        const uint64_t wgs = (segs + LC_WAVES - 1) / LC_WAVES;
        hipLaunchKernelGGL(k_load_count, dim3((unsigned)std::min<uint64_t>(wgs, (uint64_t)c->num_cus)), dim3(PC_THREADS),
                           LC_LDS_BYTES, c->stream, c->d_bytes, off, c->d_round_lb, (uint64_t)c->n_chunks, n, c->d_ids[0],
                           c->d_mat, c->vcap);
        LAUNCHCHK(c, "k_load_count");
        hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1), 0, c->stream, c->d_st, (unsigned long long)n);
        LAUNCHCHK(c, "k_init_state");
        TRY(prof_end(c));
    } else {
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
        TRY(prof_end(c));
    }
    c->apply_target = 0;
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
        } else if (c->k1 >= 3 && !c->weighted && c->vcap <= 65536) {
            // (4-byte LDS slots with 15-bit counts: unit increments, ids below 2^16)
            hipLaunchKernelGGL(k_pair_count_h32, dim3(grid_for(n, 4 * PC_THREADS, c->num_cus)),
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

SlotRefH stream_ref_h(const bpe_ctx *c) {
    SlotRefH r;
    r.b0 = c->d_ids[0];
    r.b1 = c->d_ids[1];
    r.hdr = c->d_hdr2[c->mq];
    r.T = c->slot_T;
    return r;
}

// index live: the a == b pass of a sparse iteration visits the slots k_select (sharded: k_dp_cand, once the pair is
// known) lists for it and adds the pairs it creates to the index (no "visit always" marks, no rebuild afterwards)
inline bool aa_through_index(const bpe_ctx *c) { return c->aa_sparse && c->idx_live; }

// K2 + tie-break: after this the pair is final in st (sharded streams: resolved_pair() gives
// this rank's candidate)
int launch_select(bpe_ctx *c, bool rowmax_all, bool sparse_next = false) {
    TRY(prof_begin(c, BPE_PROF_ARGMAX, 0));
    if (c->forced) {  // (encode as a replay of training: the pair is given; a sparse pass's candidate list as in the sharded loop)
        hipLaunchKernelGGL(GK(c, k_forced_pair), dim3(1), dim3(1), 0, c->stream, c->d_st, c->d_forced, (uint32_t)c->prof_iter,
                           c->d_mat, c->vcap);
        LAUNCHCHK(c, "k_forced_pair");
        if (sparse_next && c->slotted && c->slot2) {
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
        TRY(prof_end(c));
        return BPE_OK;
    }
    if (rowmax_all) {
        hipLaunchKernelGGL(k_rowmax_all, dim3(c->vcur), dim3(256), 0, c->stream, c->d_mat, c->vcap,
                           c->vcur, c->d_rowmax);
        LAUNCHCHK(c, "k_rowmax_all");
    }
    const uint64_t nslots = c->slotted ? c->slot_T : ntiles_of(c->n);
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nslots, TIE_BLOCKS));
    CandArgs C;
    C.idx = c->d_idx;
    C.dirty = c->d_idx_dirty;
    C.cand = c->d_cand;
    C.stride = (uint32_t)c->idx_cap_words;
    C.T = (uint32_t)c->slot_T;
    C.enable = sparse_next ? 1u : 0u;  // the block that makes the pair final lists the slots a sparse pass visits
    C.tie_index = (c->slotted && c->slot2 && c->idx_live && c->tie_index) ? 1u : 0u;
    C.tie_window = c->tie_window ? 1u : 0u;
    C.aa = (sparse_next && aa_through_index(c)) ? 1u : 0u;
    if (c->slotted && c->slot2)
        hipLaunchKernelGGL(GK(c, k_select<SlotRefH>), dim3(blocks), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                           c->vcap, c->vcur, c->d_st, stream_ref_h(c), c->par, c->dp_active ? 1 : 0,
                           ++c->sel_epoch, C);
    else
        hipLaunchKernelGGL(k_select<SlotRef>, dim3(blocks), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                           c->vcap, c->vcur, c->d_st, stream_ref(c), c->par, c->dp_active ? 1 : 0,
                           ++c->sel_epoch, C);
    LAUNCHCHK(c, "k_select");
    TRY(prof_end(c));
    return BPE_OK;
}

// K2 of a lean iteration while the index is live, fused with the row maxima the previous lean table
// update left to do: workgroup 0 decides (or defers to the general path), the others re-scan rows
int launch_rowsel_lean(bpe_ctx *c) {
    TRY(prof_begin(c, BPE_PROF_ARGMAX, 0));
    CandArgs C;
    C.idx = c->d_idx;
    C.dirty = c->d_idx_dirty;
    C.cand = nullptr;
    C.stride = (uint32_t)c->idx_cap_words;
    C.T = (uint32_t)c->slot_T;
    C.enable = 0;
    C.tie_index = 1;
    C.tie_window = 0;
    C.aa = 0;
    hipLaunchKernelGGL(GK(c, k_rowsel_lean), dim3(1 + (unsigned)c->lean_scan), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                       c->vcap, c->vcur, c->d_st, stream_ref_h(c), C, c->d_dbits, c->d_lean_res, ++c->lean_tag);
    LAUNCHCHK(c, "k_rowsel_lean");
    TRY(prof_end(c));
    c->rows_pending = false;
    return BPE_OK;
}

// K2 of a lean iteration right after a lean table update: from that update's per-wave records
int launch_sel_lean(bpe_ctx *c) {
    TRY(prof_begin(c, BPE_PROF_ARGMAX, 0));
    CandArgs C;
    C.idx = c->d_idx;
    C.dirty = c->d_idx_dirty;
    C.cand = nullptr;
    C.stride = (uint32_t)c->idx_cap_words;
    C.T = (uint32_t)c->slot_T;
    C.enable = 0;
    C.tie_index = 1;
    C.tie_window = 0;
    C.aa = 0;
    // the update before this selection made token vcur - 1: one record per wave of its token workgroups
    const uint32_t nwv = ((c->vcur + 255u) / 256u) * 4u;
    hipLaunchKernelGGL(GK(c, k_sel_lean), dim3(1 + (unsigned)c->lean_scan), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                       c->vcap, c->vcur, c->d_st, stream_ref_h(c), C, c->d_dbits, c->d_lean_res, ++c->lean_tag,
                       c->d_lean_sum, nwv, (uint32_t)c->lean_chain);
    LAUNCHCHK(c, "k_sel_lean");
    TRY(prof_end(c));
    c->rows_pending = false;
    return BPE_OK;
}

// the row maxima a lean table update left to do, on their own (ncols = ids in use now)
int flush_lean_rows(bpe_ctx *c, uint32_t ncols) {
    if (!c->rows_pending) return BPE_OK;
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    hipLaunchKernelGGL(GK(c, k_rowmax_lean), dim3((unsigned)c->lean_scan), dim3(1024), 0, c->stream, c->d_mat, c->vcap,
                       c->d_rowmax, c->d_st, ncols, c->d_dbits);
    LAUNCHCHK(c, "k_rowmax_lean");
    TRY(prof_end(c));
    c->rows_pending = false;
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
    if (nt > (uint64_t)c->scan_sup_min) {  // many tiles: three small launches instead of one workgroup's serial walk
        static_assert(sizeof(TS) == 24, "a TS is three 8-byte words");
        const uint64_t nsup = (nt + SUP_TILES - 1) / SUP_TILES;
        TS *ssum = reinterpret_cast<TS *>(c->d_sup), *spre = ssum + (nt / SUP_TILES + 2);
        hipLaunchKernelGGL(k_tile_sup, dim3((unsigned)nsup), dim3(SUP_TILES), 0, c->stream, c->d_tsum, nt, c->d_st, c->par, ssum);
        hipLaunchKernelGGL(k_tile_scan_sup, dim3(1), dim3(1024), 0, c->stream, ssum, nsup, spre, c->d_st, c->par, rec, iter,
                           c->d_ids[c->par], c->d_dirty_n);
        hipLaunchKernelGGL(k_tile_expand, dim3((unsigned)nsup), dim3(SUP_TILES), 0, c->stream, c->d_tsum, nt, c->d_st, c->par,
                           spre, c->d_tile_off, c->d_tile_sin);
    } else {
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, c->stream, c->d_tsum, nt, c->d_tile_off,
                           c->d_tile_sin, c->d_st, c->par, rec, iter, c->d_ids[c->par], c->d_dirty_n);
    }
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

// ---- slotted stream, second form (k_slots2.hip) -------------------------------------------------
constexpr unsigned SPARSE_GRID = 1024;      // resident workgroups of a sparse pass (4 per CU, four waves = four slots each)

int index_build(bpe_ctx *c) {
    const uint64_t nwords = (c->slot_T + 31) / 32;
    // (rows = buckets of the geometry the stream is in: 32 Ki for 1024-id slots, 8 Ki for 256-id slots -- which has four
    // times the slots, hence four times the words per row: the same bytes either way)
    if (nwords + 4 > c->idx_cap_words || idx_h(c) > c->idx_cap_rows) {
        if (c->d_idx) HIPCHK(c, hipFree(c->d_idx));
        if (c->d_idx_dirty) HIPCHK(c, hipFree(c->d_idx_dirty));
        c->d_idx = c->d_idx_dirty = nullptr;
        const uint64_t cap = (nwords + 4 + 63) / 64 * 64;  // row stride: 16-byte aligned rows, a padded tail
        if (c->d_idx_tmp) HIPCHK(c, hipFree(c->d_idx_tmp));
        c->d_idx_tmp = nullptr;
        c->idx_cap_words = c->idx_cap_rows = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_idx, cap * idx_h(c) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void **)&c->d_idx_tmp, cap * idx_h(c) * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void **)&c->d_idx_dirty, cap * sizeof(uint32_t)));
        c->idx_cap_words = cap;
        c->idx_cap_rows = idx_h(c);
    }
    if (nwords) {
        hipLaunchKernelGGL(GK(c, k_index_build), dim3((unsigned)nwords), dim3(1024), (size_t)idx_h(c) * 4, c->stream,
                           c->d_ids[0], c->d_ids[1], c->d_hdr2[c->mq], (uint32_t)c->slot_T, c->d_idx_tmp,
                           c->d_idx_dirty, c->d_st);
        LAUNCHCHK(c, "k_index_build");
        hipLaunchKernelGGL(GK(c, k_index_transpose), dim3((unsigned)((nwords + 31) / 32), idx_h(c) / 32), dim3(256), 0,
                           c->stream, c->d_idx_tmp, (uint32_t)nwords, c->d_idx, (uint32_t)c->idx_cap_words);
        LAUNCHCHK(c, "k_index_transpose");
    }
    c->idx_live = true;
    c->idx_rebuild = false;
    c->n_index_builds++;
    return BPE_OK;
}

int slots2_enter(bpe_ctx *c) {
    c->slot_T = (c->n + (uint64_t)c->ts - 1) / (uint64_t)c->ts;
    c->mq = 0;
    hipLaunchKernelGGL(GK(c, k_slot2_init), dim3(grid_for(std::max<uint64_t>(c->slot_T, 1), 256, c->num_cus * 4)),
                       dim3(256), 0, c->stream, c->d_hdr2[0], c->slot_T, c->d_st, c->par, (uint32_t)c->par,
                       c->d_ids[c->par]);
    LAUNCHCHK(c, "k_slot2_init");
    c->slotted = true;
    c->slot2 = true;
    // (a pass that was cut short by a device status may have left staged-header marks behind)
    HIPCHK(c, hipMemsetAsync(c->d_smask, 0, (c->slot_T / 32 + 1) * sizeof(uint32_t), c->stream));
    if (c->idx_live) TRY(index_build(c));  // the slots changed: the index is rebuilt
    return BPE_OK;
}

// slots -> contiguous in d_ids[0] (par 0); st->n[0] = the stream length
int slots2_leave(bpe_ctx *c) {
    const uint64_t T = c->slot_T;
    if (T) {
        const uint64_t nb = (T + SCAN_TILE - 1) / SCAN_TILE;
        hipLaunchKernelGGL(GK(c, k_slot2_lens), dim3(grid_for(T, 256, c->num_cus * 4)), dim3(256), 0, c->stream,
                           c->d_hdr2[c->mq], T, c->d_slot_lens);
        hipLaunchKernelGGL(k_scan_blocksum, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_slot_lens, T,
                           c->d_slot_bsum);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->d_slot_bsum, nb,
                           c->d_scratch + 3);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, c->d_slot_lens, T,
                           c->d_slot_bsum, c->d_slot_off);
        hipLaunchKernelGGL(GK(c, k_slot2_compact), dim3((unsigned)((T + 3) / 4)), dim3(256), 0, c->stream, c->d_ids[0],
                           c->d_ids[1], c->d_hdr2[c->mq], T, c->d_slot_off, c->d_ids2);
        LAUNCHCHK(c, "k_slot2_compact");
    }
    std::swap(c->d_ids[0], c->d_ids2);
    if (c->par != 0) {
        hipLaunchKernelGGL(k_move_n, dim3(1), dim3(1), 0, c->stream, c->d_st, c->par, 0);
        LAUNCHCHK(c, "k_move_n");
    }
    c->par = 0;
    c->slotted = false;
    c->slot2 = false;
    return BPE_OK;
}

// replicas of the delta vectors for this pass: as many as the buffer holds at this vocabulary
// (the hot tokens of an early merge serialise at ~11 ns per same-address atomic), fewer when
// the merged pair is rare (the table update folds every replica).  dstride = vector stride.
inline uint32_t delta_layout(const bpe_ctx *c, uint32_t Z) {
    const uint32_t dstride = std::min<uint32_t>(c->vcap, ((Z + 1 + 63) / 64) * 64);
    const uint64_t buf_words = (uint64_t)c->vcap * 4 * DELTA_REPL;
    int shift = 0;
    while (shift < 7 && ((uint64_t)dstride * 4 << (shift + 1)) <= buf_words) shift++;  // (the skew has its own room; a single pair's
                                                                                          // pass uses at most 128 of the DELTA_REPL blocks)
    const uint64_t cnt = c->last_count;
    const int want = cnt >= (1u << 20) ? 8 : (cnt >= (1u << 16) ? 5 : (cnt >= (1u << 12) ? 3 : 0));
    return dstride | ((uint32_t)std::min(std::min(shift, std::max(want, c->rep_min)), c->rep_max) << 24);
}

// Will this iteration's a != b pass be a sparse one?  It pays when the pair is rare enough that
// most slots cannot hold it.  Decided (and the index brought up to date) BEFORE k_select, whose
// deciding block makes the pass's candidate list.  (use_sparse == 2: every a != b pass is a
// sparse one -- tests drive the sparse kernel and the index through streams of a few slots.)
int plan_pass2(bpe_ctx *c, bool *sparse_out) {
    const uint32_t T = (uint32_t)c->slot_T;
    // A stream of a few thousand slots: visiting them all costs nothing, an iteration is launches and round
    // trips only -- and the index is what lets a lean iteration select from the table update's records and
    // chain tied merges (k_sel_lean): indexed as soon as the iterations are lean ones (measured on the
    // de-duplicated 1 GB input, 1127 slots: 27.7 k -> 48.2 k merges/s)
    const bool small = T <= 16 * SPARSE_GRID;
    const bool can_index = c->use_sparse && T > 0 && (c->use_sparse == 2 || !small || c->lean);
    const bool rare = c->last_count != ~0ull && (small ? c->last_count <= (uint64_t)c->lean_count
                                                       : c->last_count * (uint64_t)c->sparse_ratio < T);
    const bool sparse = can_index && (c->use_sparse == 2 || rare);
    // (an a == b pass only marks the slots it rewrote as "visit always": once the host has seen
    // one go by, the index is rebuilt so that those marks do not pile up)
    if (sparse && c->small_slots && c->ts != TILE2_MIN && (!small || c->small_slots == 2)) {
        // a large stream goes sparse: from here on a merge site costs the slot it sits in -- re-pack into 256-id slots
        // (kernels of namespace bpe_g1 from the next launch on; the index is built over the new slots -- below, or by
        // slots2_enter itself when the sharded loop had one already, for its ties)
        TRY(slots2_leave(c));
        c->ts = TILE2_MIN;
        TRY(slots2_enter(c));
    }
    if (sparse && (!c->idx_live || c->idx_rebuild)) TRY(index_build(c));
    *sparse_out = sparse;
    return BPE_OK;
}

// one merge pass of the second slotted form + table update.  The host does not know the pair
// (it runs `depth` merges ahead): the a != b kernel and the a == b kernel are both launched and
// the one the pair does not call for returns at once.
int launch_passes2(bpe_ctx *c, uint32_t newid, bool sparse, uint32_t dl) {
    TRY(prof_begin(c, BPE_PROF_MERGE, 0));
    if ((++c->epoch & EPOCH_MASK) == 0) {  // tag wrapped: retire every old descriptor
        HIPCHK(c, hipMemsetAsync(c->d_desc, 0, ((TILE / TILE2_MIN) * c->cap_tiles + 4) * sizeof(unsigned long long), c->stream));
        c->epoch++;
    }
    const uint32_t T = (uint32_t)c->slot_T;
    AbArgs A;
    A.b0 = c->d_ids[0];
    A.b1 = c->d_ids[1];
    A.hdr_in = c->d_hdr2[c->mq];
    A.hdr_out = c->d_hdr2[c->mq ^ 1];
    A.stage = c->d_stage;
    A.smask = c->d_smask;
    A.T = T;
    A.st = c->d_st;
    A.newid = newid;
    A.delta = c->exp_no_delta ? nullptr : c->d_delta;
    A.vcap = dl;
    A.idx = c->idx_live ? c->d_idx : nullptr;
    A.istride = (uint32_t)c->idx_cap_words;
    A.cand = c->d_cand;
    A.removed = c->d_removed;
    A.dirty_n = c->d_dirty_n;
    // every id the pass can meet is below newid: small enough for the LDS delta tables?
    const bool ldsd = c->lds_delta && newid + 1 <= (uint32_t)LDSD_CAP;
    if (sparse) {
        if (ldsd)
            hipLaunchKernelGGL(GK(c, k_merge_ab_sparse<true>), dim3(SPARSE_GRID), dim3(MT), 0, c->stream, A);
        else
            hipLaunchKernelGGL(GK(c, k_merge_ab_sparse<false>), dim3(SPARSE_GRID), dim3(MT), 0, c->stream, A);
        c->n_sparse++;
    } else {
        const unsigned g = std::max((T + MT / 64 - 1) / (MT / 64), 1u);  // one wave per slot
        const unsigned ge = std::min(g, 5u * (unsigned)c->num_cus);       // ... or a resident grid
        if (ldsd && A.idx)
            hipLaunchKernelGGL(GK(c, k_merge_ab_dense_early<true>), dim3(ge), dim3(MT), 0, c->stream, A);
        else if (ldsd)
            hipLaunchKernelGGL(GK(c, k_merge_ab_dense_early<false>), dim3(ge), dim3(MT), 0, c->stream, A);
        else if (A.idx)
            hipLaunchKernelGGL(GK(c, k_merge_ab_dense<true>), dim3(g), dim3(MT), 0, c->stream, A);
        else
            hipLaunchKernelGGL(GK(c, k_merge_ab_dense<false>), dim3(g), dim3(MT), 0, c->stream, A);
        c->n_dense++;
    }
    LAUNCHCHK(c, "k_merge_ab");
    AaArgs B;
    B.b0 = c->d_ids[0];
    B.b1 = c->d_ids[1];
    B.w0 = c->d_ids[0];
    B.w1 = c->d_ids[1];
    B.hdr_in = A.hdr_in;
    B.hdr_out = A.hdr_out;
    B.stage = sparse ? c->d_stage : nullptr;
    B.smask = c->d_smask;
    B.T = T;
    B.st = c->d_st;
    B.newid = newid;
    B.delta = c->d_delta;
    B.vcap = dl;
    B.sdesc = c->d_desc;
    B.epoch = c->epoch;
    const bool aas = sparse && aa_through_index(c);
    B.dirty = (c->idx_live && !aas) ? c->d_idx_dirty : nullptr;
    B.removed = c->d_removed;
    B.cand = aas ? c->d_cand : nullptr;
    B.idx = aas ? c->d_idx : nullptr;
    B.istride = (uint32_t)c->idx_cap_words;
    c->last_aa_indexed = aas;
    // (a resident grid of single-wave workgroups -- a slot may wait for its predecessor's carry:
    // ~140 VGPRs admit three waves per SIMD, twelve per CU; eight are launched)
    hipLaunchKernelGGL(GK(c, k_merge_aa), dim3(std::max(1u, std::min(T, 8u * (unsigned)c->num_cus))), dim3(64), 0,
                       c->stream, B);
    LAUNCHCHK(c, "k_merge_aa");
    TRY(prof_end(c));
    return BPE_OK;
}

// table update of the second slotted form (+ commit of staged headers, stream length, record);
// folded: the delta comes from the all-reduced payload of sharded training
int launch_table2(bpe_ctx *c, uint32_t newid, int iter, IterRec *rec, bool sparse, uint32_t dl, bool folded) {
    const uint32_t T = (uint32_t)c->slot_T;
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    const uint32_t na = (newid + 1 + 31) / 32;
    if (folded)
        hipLaunchKernelGGL(k_apply2<true>, dim3(na + 8), dim3(256), 0, c->stream, c->d_mat, c->vcap, c->d_dp_folded,
                           c->vcap, c->d_rowmax, c->d_st, newid, c->d_dirty_list, c->d_dirty_n, c->par, rec, iter,
                           na, c->d_hdr2[c->mq], c->d_stage, c->d_removed, c->d_smask, (T + 31) / 32);
    else
        hipLaunchKernelGGL(k_apply2<false>, dim3(na + 8), dim3(256), 0, c->stream, c->d_mat, c->vcap, c->d_delta, dl,
                           c->d_rowmax, c->d_st, newid, c->d_dirty_list, c->d_dirty_n, c->par, rec, iter, na,
                           c->d_hdr2[c->mq], c->d_stage, c->d_removed, c->d_smask, (T + 31) / 32);
    LAUNCHCHK(c, "k_apply2");
    hipLaunchKernelGGL(k_rowmax_list, dim3(ROW_BLOCKS), dim3(1024), 0, c->stream, c->d_mat, c->vcap, newid + 1,
                       c->d_rowmax, c->d_st, c->d_dirty_list, c->d_dirty_n);
    LAUNCHCHK(c, "k_rowmax_list");
    TRY(prof_end(c));
    c->par ^= 1;
    if (!sparse) c->mq ^= 1;
    c->stats_valid = false;
    c->stream_is_bytes = false;
    return BPE_OK;
}

int launch_merge2(bpe_ctx *c, uint32_t newid, int iter, IterRec *rec, bool sparse) {
    const uint32_t dl = delta_layout(c, newid);
    TRY(launch_passes2(c, newid, sparse, dl));
    return launch_table2(c, newid, iter, rec, sparse, dl, false);
}

// A lean iteration (k_lean.hip) after its selection: the merge pass (waves find their own candidates),
// then the table update -- which leaves the row maxima to the next launch that needs them
// (k_rowsel_lean, or flush_lean_rows).  use_index: candidates come from the inverted index (it is
// live); otherwise every live slot is visited.
int launch_lean(bpe_ctx *c, uint32_t newid, int iter, IterRec *rec, bool use_index) {
    const uint32_t T = (uint32_t)c->slot_T;
    const uint32_t dl = delta_layout(c, newid);
    AbArgs A;
    A.b0 = c->d_ids[0];
    A.b1 = c->d_ids[1];
    A.hdr_in = c->d_hdr2[c->mq];
    A.hdr_out = nullptr;
    A.stage = c->d_stage;
    A.smask = c->d_smask;
    A.T = T;
    A.st = c->d_st;
    A.newid = newid;
    A.delta = c->d_delta;
    A.vcap = dl;
    A.idx = c->idx_live ? c->d_idx : nullptr;
    A.istride = (uint32_t)c->idx_cap_words;
    A.cand = nullptr;
    A.removed = c->d_removed;
    A.dirty_n = c->d_dirty_n;
    const uint32_t nwords = (T + 31) / 32;
    // a resident grid: one 1024-thread workgroup per CU at most; a workgroup takes at least one mask word
    // (index: 32 slots, two per wave -- a stream of 40 k slots must not end up on a third of the CUs: cfg2's
    // lean passes were 1.4x slower than the sparse kernel they replaced while a workgroup took 16 words) or
    // 16 slots (no index: one per wave)
    const unsigned g = std::max(1u, std::min(use_index ? nwords : (T + 15) / 16, (unsigned)c->lean_grid));
    TRY(prof_begin(c, BPE_PROF_MERGE, 0));
    if (c->idx_live)
        hipLaunchKernelGGL(GK(c, k_merge_ab_lean<true>), dim3(g), dim3(LEAN_MT), 0, c->stream, A, c->d_idx_dirty,
                           use_index ? 1u : 0u, c->d_dbits);
    else
        hipLaunchKernelGGL(GK(c, k_merge_ab_lean<false>), dim3(g), dim3(LEAN_MT), 0, c->stream, A, (const uint32_t *)nullptr, 0u,
                           c->d_dbits);
    LAUNCHCHK(c, "k_merge_ab_lean");
    TRY(prof_end(c));
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    const uint32_t na = (newid + 1 + 255) / 256;
    // (+ workgroups that commit the staged headers: a mask word or two per thread)
    const uint32_t ncommit = std::max(8u, std::min(128u, (nwords + 255) / 256));  // (a mask word per thread, at most)
    hipLaunchKernelGGL(GK(c, k_apply_lean), dim3(na + ncommit), dim3(256), 0, c->stream, c->d_mat, c->vcap, c->d_delta, dl,
                       c->d_rowmax, c->d_st, newid, c->d_dbits, c->par, rec, iter, na, c->d_hdr2[c->mq], c->d_stage,
                       c->d_removed, c->d_smask, nwords, c->d_lean_sum);
    LAUNCHCHK(c, "k_apply_lean");
    TRY(prof_end(c));
    c->par ^= 1;  // (a sparse-style pass: staged headers, the header arrays do not flip)
    c->stats_valid = false;
    c->stream_is_bytes = false;
    c->rows_pending = true;
    c->n_lean++;
    if (use_index) c->n_sparse++; else c->n_dense++;
    return BPE_OK;
}

// A chain step (k_chain.hip): selection or list look-ups, one merge pass for the whole batch, table update.
// zhi: the largest token id the step can make (the device counts the merges; the host only knows a bound).
// records: the last launch that touched the pair table was a k_apply_chain.
int launch_chain_step(bpe_ctx *c, uint32_t step, uint32_t zhi, bool use_index, bool records, bool dense = false) {
    const uint32_t T = (uint32_t)c->slot_T;
    const uint32_t dl = delta_layout(c, zhi);
    // one launch for the whole step (k_step.hip) where there is nothing between its parts: a single-GPU job's sparse steps
    const bool fused = c->fuse_step && !c->forced && !dense && !c->dp_comm && use_index && c->idx_live;
    if (!fused) TRY(prof_begin(c, BPE_PROF_ARGMAX, 0));
    CandArgs C;
    C.idx = c->d_idx;
    C.dirty = c->d_idx_dirty;
    C.cand = nullptr;
    C.stride = (uint32_t)c->idx_cap_words;
    C.T = T;
    C.enable = 0;
    C.tie_index = 1;
    C.tie_window = 0;
    C.aa = 0;
    if (!c->idx_live) C.T = 0;  // (no index: a tie finds none of its pairs through it, and the step defers)
    // (sharded dense steps: some ranks may hold an index, others not -- a tie ordered by part of the ranks would be
    // ordered wrongly: nobody orders it, every rank reports "no occurrence" and the general path decides)
    if (dense && c->dp_comm) C.T = 0;
    (void)records;
    if (!c->d_chain_req) {
        HIPCHK(c, hipMalloc((void **)&c->d_chain_req, PL_REQ_WORDS * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->d_chain_req, 0, PL_REQ_WORDS * sizeof(unsigned long long), c->stream));
    }
    if (!c->d_pool) {
        HIPCHK(c, hipMalloc((void **)&c->d_pool, (3 * PL_CAP + 128) * sizeof(PoolEnt)));  // (+ a sharded selection's hand-over: k_pool_sel -> k_pool_sel_dp)
        HIPCHK(c, hipMalloc((void **)&c->d_pool_gather, (2 + 2 * PL_GATHER) * sizeof(uint32_t)));
        HIPCHK(c, hipMemsetAsync(c->d_pool, 0, (3 * PL_CAP + 128) * sizeof(PoolEnt), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pool_gather, 0, (2 + 2 * PL_GATHER) * sizeof(uint32_t), c->stream));
    }
    // sharded training (dp_train_loop, api_rccl.hip): the step's two collectives sit between its launches -- a tie's
    // first occurrences (MIN) before the batch is formed, the batch's delta (SUM) before the table update: k_pool_sel
    // leaves the local first occurrences of the entries it must order, k_pool_sel_dp finishes the selection from the
    // reduced words (the pool itself is replicated state, maintained alike on every rank)
    const DpComm *dp = c->dp_comm;
    uint32_t kcap = (uint32_t)(dense ? CH_KDENSE : std::min(CH_KSWEEP, c->chain_kcap));
    if (dp) kcap = std::min(kcap, (uint32_t)c->dp_kcap);
    const uint32_t hint_below = (uint32_t)(c->pool_hint > 0 ? c->pool_hint : (int)kcap);
    // (the deciding workgroup and the scanning ones wait for each other: all of them must be resident at once -- one
    // 1024-thread workgroup per CU at most, like lean_grid)
    const unsigned nscan = (unsigned)std::max(1, std::min(c->chain_scan, c->num_cus - 1));
    if (fused) {
        // ---- the whole step as ONE launch (k_step.hip) ----------------------------------------------------------------
        if (!c->d_step_pub) {
            HIPCHK(c, hipMalloc((void **)&c->d_step_pub, 64 * sizeof(unsigned long long)));
            HIPCHK(c, hipMemsetAsync(c->d_step_pub, 0, 64 * sizeof(unsigned long long), c->stream));
        }
        if (!c->d_step_bar) {
            HIPCHK(c, hipMalloc((void **)&c->d_step_bar, 256));
            HIPCHK(c, hipMemsetAsync(c->d_step_bar, 0, 256, c->stream));
            c->step_bar_target = 0;
        }
        const uint32_t nwords = (T + 31) / 32;
        StepArgs S;
        S.rowmax = c->d_rowmax;
        S.mat = c->d_mat;
        S.stride = c->vcap;
        S.st = c->d_st;
        S.ref = stream_ref_h(c);
        S.C = C;
        S.dbits = c->d_dbits;
        S.res = c->d_lean_res;
        S.tag = ++c->lean_tag;
        S.req = c->d_chain_req;
        S.kcap = kcap;
        S.pool = c->d_pool;
        S.gather = c->d_pool_gather;
        S.hint_below = hint_below;
        AbArgs &A = S.A;
        A.b0 = c->d_ids[0];
        A.b1 = c->d_ids[1];
        A.hdr_in = c->d_hdr2[c->mq];
        A.hdr_out = nullptr;
        A.stage = c->d_stage;
        A.smask = c->d_smask;
        A.T = T;
        A.st = c->d_st;
        A.newid = 0;  // (the device knows: the published line)
        A.delta = c->d_delta;
        A.vcap = dl;
        A.idx = c->d_idx;
        A.istride = (uint32_t)c->idx_cap_words;
        A.cand = nullptr;
        uint32_t *const removed = (c->weighted || !c->count_is_removed) ? c->d_removed : nullptr;
        A.removed = removed;
        A.dirty_n = c->d_dirty_n;
        S.idx_dirty = c->d_idx_dirty;
        S.use_index = 1u | (c->chain_prefetch ? 2u : 0u);
        // the grid: what the merge pass wants, what the table update needs (a token per thread), the scanning workgroups --
        // all of it resident at once: one 1024-thread workgroup per CU at most
        const unsigned gapply = (zhi + 1 + 1023) / 1024;
        const unsigned cap = std::min((unsigned)c->num_cus, std::max(std::max((unsigned)c->lean_grid, gapply), 2u));
        const unsigned gmerge = std::max(1u, std::min(std::min(nwords, (unsigned)c->lean_grid), cap));
        const unsigned ns = std::min(nscan, cap - 1);
        const unsigned G = std::min(cap, std::max(std::max(gmerge, gapply), 1 + ns));
        if (G < gapply) return fail(c, BPE_E_INTERNAL, "fused chain step: %u workgroups cannot hold %u tokens", G, zhi + 1);
        S.nscan = ns;
        S.gm = gmerge;
        S.delta = c->d_delta;
        S.dl = dl;
        S.par = c->par;
        S.rec = c->h_rec;
        S.srec = c->h_srec;
        S.step = step;
        S.hdr_cur = c->d_hdr2[c->mq];
        S.stage = c->d_stage;
        S.removed = removed;
        S.smask = c->d_smask;
        S.nwords = nwords;
        S.sums = c->d_lean_sum;
        S.pub = c->d_step_pub;
        S.bar = c->d_step_bar;
        c->step_bar_target += G;
        S.bar_target = c->step_bar_target;
        S.stamps = c->d_step_stamps;
        TRY(prof_begin(c, BPE_PROF_MERGE, 0));
        hipLaunchKernelGGL(GK(c, k_step), dim3(G), dim3(LEAN_MT), 0, c->stream, S);
        LAUNCHCHK(c, "k_step");
        TRY(prof_end(c));
        c->par ^= 1;
        c->stats_valid = false;
        c->stream_is_bytes = false;
        c->rows_pending = true;
        c->n_steps++;
        c->n_fused++;
        c->n_sparse++;
        return BPE_OK;
    }
    if (c->forced)
        hipLaunchKernelGGL(GK(c, k_forced_sel), dim3(1), dim3(64), 0, c->stream, c->d_st, c->d_forced, c->d_mat, c->vcap, kcap);
    else
        hipLaunchKernelGGL(GK(c, k_pool_sel), dim3(1 + nscan), dim3(1024), 0, c->stream, c->d_rowmax, c->d_mat,
                           c->vcap, c->d_st, stream_ref_h(c), C, c->d_dbits, c->d_lean_res, ++c->lean_tag, c->d_chain_req,
                           kcap, c->d_pool, c->d_pool_gather, hint_below, dp ? c->d_dp_ckey : (long long *)nullptr,
                           (unsigned long long)(dp ? dp->rank : 0), c->d_pool + PL_CAP);
    LAUNCHCHK(c, "k_pool_sel");
    if (dp) {
        TRY(dp_allreduce(c, c->d_dp_ckey, DP_KEY_WORDS, BPE_DT_INT64, BPE_OP_MIN));
        hipLaunchKernelGGL(GK(c, k_pool_sel_dp), dim3(1), dim3(PL_CAP), 0, c->stream, c->d_st, c->d_dp_ckey, kcap, c->d_pool,
                           c->d_pool + PL_CAP, hint_below);
        LAUNCHCHK(c, "k_pool_sel_dp");
    }
    TRY(prof_end(c));
    AbArgs A;
    A.b0 = c->d_ids[0];
    A.b1 = c->d_ids[1];
    A.hdr_in = c->d_hdr2[c->mq];
    A.hdr_out = dense ? c->d_hdr2[c->mq ^ 1] : nullptr;
    A.stage = c->d_stage;
    A.smask = c->d_smask;
    A.T = T;
    A.st = c->d_st;
    A.newid = 0;  // (the device knows: st->bz0)
    A.delta = c->d_delta;
    A.vcap = dl;
    A.idx = c->idx_live ? c->d_idx : nullptr;
    A.istride = (uint32_t)c->idx_cap_words;
    A.cand = nullptr;
    // (an unweighted stream: a != b merges remove exactly `count` ids -- no removal counters, one atomic less per site;
    // sharded steps keep them: the count is the GLOBAL one, the ids removed are this shard's)
    uint32_t *const removed = (c->weighted || dp || !c->count_is_removed) ? c->d_removed : nullptr;
    A.removed = removed;
    A.dirty_n = c->d_dirty_n;
    const uint32_t nwords = (T + 31) / 32;
    TRY(prof_begin(c, BPE_PROF_MERGE, 0));
    if (dense) {
        // every slot, one 1024-thread workgroup per CU at most (sixteen slots in flight each)
        const unsigned g = std::max(1u, std::min((T + 15) / 16, (unsigned)c->num_cus));
        hipLaunchKernelGGL(GK(c, k_merge_chain_dense), dim3(g), dim3(LEAN_MT), 0, c->stream, A, c->d_dbits);
        // (the batch of one has its own kernel -- five 256-thread workgroups per CU, 96 VGPRs: the other one returns at once)
        const unsigned g1 = std::max(1u, std::min((T + MT / 64 - 1) / (MT / 64), 5u * (unsigned)c->num_cus));
        hipLaunchKernelGGL(GK(c, k_merge_chain_dense1), dim3(g1), dim3(MT), 0, c->stream, A);
    } else {
        const unsigned g = std::max(1u, std::min(use_index ? nwords : (T + 15) / 16, (unsigned)c->lean_grid));
        hipLaunchKernelGGL(GK(c, k_merge_chain), dim3(g), dim3(LEAN_MT), 0, c->stream, A, c->d_idx_dirty,
                           (use_index ? 1u : 0u) | (c->chain_prefetch ? 2u : 0u), c->d_dbits);
    }
    LAUNCHCHK(c, "k_merge_chain");
    TRY(prof_end(c));
    TRY(prof_begin(c, BPE_PROF_TABLE, 0));
    const uint32_t na = (zhi + 1 + 255) / 256;
    const uint32_t ncommit = std::max(8u, std::min(128u, (nwords + 255) / 256));  // (a mask word per thread, at most)
    const uint32_t fS = std::min<uint32_t>(c->vcap, ((zhi + 1 + 63) / 64) * 64);  // vector stride of the SUM payload
    uint32_t *ftail = dp ? c->d_dp_cfold + (size_t)2 * kcap * fS : nullptr;
    if (dp) {
        hipLaunchKernelGGL(GK(c, k_dp_fold_chain), dim3(na), dim3(256), 0, c->stream, c->d_delta, dl, c->d_st, c->d_dp_cfold, fS, ftail);
        LAUNCHCHK(c, "k_dp_fold_chain");
        TRY(dp_allreduce(c, c->d_dp_cfold, (uint64_t)2 * kcap * fS + 64, BPE_DT_INT32, BPE_OP_SUM));
    }
    hipLaunchKernelGGL(GK(c, k_apply_chain), dim3(na + ncommit), dim3(256), 0, c->stream, c->d_mat, c->vcap, c->d_delta, dl, c->d_rowmax,
                       c->d_st, c->d_dbits, c->par, c->h_rec, c->h_srec, step, na, c->d_hdr2[c->mq], c->d_stage,
                       removed, c->d_smask, nwords, c->d_lean_sum, dp ? c->d_dp_cfold : (const uint32_t *)nullptr, fS,
                       (const uint32_t *)ftail);
    LAUNCHCHK(c, "k_apply_chain");
    TRY(prof_end(c));
    c->par ^= 1;
    if (dense) c->mq ^= 1;  // (a sparse step stages its headers: the header arrays do not flip)
    c->stats_valid = false;
    c->stream_is_bytes = false;
    c->rows_pending = true;
    c->n_steps++;
    if (use_index) c->n_sparse++; else c->n_dense++;
    return BPE_OK;
}

// A large buffer of the caller's (pageable memory) to the device.  hipMemcpy from pageable memory stages through the
// runtime's own pinned buffer with ONE host thread: 0.07 - 0.57 s per GB depending on the box (profiles/r5_notes.md) --
// for the 1 GB headline input (1 GB of text + 1.4 GB of chunk offsets) a fifth of a train() call.  Here STAGE_T host
// threads copy piece i into pinned buffer i % STAGE_R (each its slice) while the copy engine drains the pieces before:
// the upload runs at min(host memory bandwidth, PCIe).  The copy is complete on the stream when the call returns
// BPE_OK only in the sense of hipMemcpyAsync: the caller synchronises the stream (load_bytes_impl does).
constexpr size_t STAGE_PIECE = 32u << 20;
constexpr int STAGE_R = 4, STAGE_T_MAX = 16;
int upload_h2d(bpe_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!bytes) return BPE_OK;
    if (!c->pinned_upload || bytes < 2 * STAGE_PIECE) {
        HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return BPE_OK;
    }
    if (!c->h_stage) {
        if (hipHostMalloc((void **)&c->h_stage, STAGE_PIECE * STAGE_R, hipHostMallocDefault) != hipSuccess) {
            c->h_stage = nullptr;
            (void)hipGetLastError();
            HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));  // (no pinned memory to be had: the plain path)
            return BPE_OK;
        }
        for (int r = 0; r < STAGE_R; r++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage[r], hipEventDisableTiming));
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int T = (int)std::max(1u, std::min((unsigned)STAGE_T_MAX, hw / 2));
    const size_t pieces = (bytes + STAGE_PIECE - 1) / STAGE_PIECE;
    std::atomic<long long> go{-1};
    std::atomic<unsigned long long> done{0};
    const uint8_t *s8 = static_cast<const uint8_t *>(src);
    uint8_t *stage = c->h_stage;
    auto worker = [&](int w) {
        for (size_t i = 0; i < pieces; i++) {
            while (go.load(std::memory_order_acquire) < (long long)i) std::this_thread::yield();
            const size_t len = std::min(STAGE_PIECE, bytes - i * STAGE_PIECE);
            const size_t per = (len + (size_t)T - 1) / (size_t)T, lo = std::min(len, per * (size_t)w), hi = std::min(len, lo + per);
            if (hi > lo) memcpy(stage + (i % STAGE_R) * STAGE_PIECE + lo, s8 + i * STAGE_PIECE + lo, hi - lo);
            done.fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)T);
    for (int w = 1; w < T; w++) pool.emplace_back(worker, w);
    int rc = BPE_OK;
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < pieces; i++) {
        // (the buffer's previous piece has left it -- this call's, or the tail of the call before: a copy is only ENQUEUED
        // when a call returns; on an error the workers are still let through every piece, so that they end)
        if (c->stage_busy[i % STAGE_R] && err == hipSuccess) {
            err = hipEventSynchronize(c->ev_stage[i % STAGE_R]);
            c->stage_busy[i % STAGE_R] = false;
        }
        go.store((long long)i, std::memory_order_release);
        {   // the calling thread is worker 0
            const size_t len = std::min(STAGE_PIECE, bytes - i * STAGE_PIECE);
            const size_t per = (len + (size_t)T - 1) / (size_t)T, hi = std::min(len, per);
            if (hi) memcpy(stage + (i % STAGE_R) * STAGE_PIECE, s8 + i * STAGE_PIECE, hi);
            done.fetch_add(1, std::memory_order_release);
        }
        while (done.load(std::memory_order_acquire) < (unsigned long long)T * (i + 1)) std::this_thread::yield();
        if (err == hipSuccess) {
            const size_t len = std::min(STAGE_PIECE, bytes - i * STAGE_PIECE);
            err = hipMemcpyAsync(static_cast<uint8_t *>(dst) + i * STAGE_PIECE, stage + (i % STAGE_R) * STAGE_PIECE, len,
                                 hipMemcpyHostToDevice, c->stream);
            if (err == hipSuccess) err = hipEventRecord(c->ev_stage[i % STAGE_R], c->stream);
            if (err == hipSuccess) c->stage_busy[i % STAGE_R] = true;
        }
    }
    for (std::thread &t : pool) t.join();
    if (err != hipSuccess) rc = fail(c, BPE_E_HIP, "pinned upload: %s", hipGetErrorString(err));
    return rc;
}

// ... and the way back: the copy engine fills pinned buffer i % STAGE_R with piece i (up to STAGE_R pieces ahead) while
// the host threads copy the pieces before out into the caller's (pageable) buffer.  Returns with the data in dst.
int download_d2h(bpe_ctx *c, void *dst, const void *src_dev, size_t bytes) {
    if (!bytes) return BPE_OK;
    if (!c->pinned_upload || bytes < 2 * STAGE_PIECE || !c->h_stage) {  // (the ring exists once an upload has used it)
        HIPCHK(c, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BPE_OK;
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int T = (int)std::max(1u, std::min((unsigned)STAGE_T_MAX, hw / 2));
    const size_t pieces = (bytes + STAGE_PIECE - 1) / STAGE_PIECE;
    std::atomic<long long> go{-1};
    std::atomic<unsigned long long> done{0};
    uint8_t *d8 = static_cast<uint8_t *>(dst);
    const uint8_t *stage = c->h_stage;
    auto copy_out = [&](size_t i, int w) {
        const size_t len = std::min(STAGE_PIECE, bytes - i * STAGE_PIECE);
        const size_t per = (len + (size_t)T - 1) / (size_t)T, lo = std::min(len, per * (size_t)w), hi = std::min(len, lo + per);
        if (hi > lo) memcpy(d8 + i * STAGE_PIECE + lo, stage + (i % STAGE_R) * STAGE_PIECE + lo, hi - lo);
        done.fetch_add(1, std::memory_order_release);
    };
    auto worker = [&](int w) {
        for (size_t i = 0; i < pieces; i++) {
            while (go.load(std::memory_order_acquire) < (long long)i) std::this_thread::yield();
            copy_out(i, w);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)T);
    for (int w = 1; w < T; w++) pool.emplace_back(worker, w);
    hipError_t err = hipSuccess;
    size_t issued = 0;
    for (size_t i = 0; i < pieces; i++) {
        for (; issued < pieces && issued < i + (size_t)STAGE_R && err == hipSuccess; issued++) {  // (pieces before i are out of their buffers)
            const size_t len = std::min(STAGE_PIECE, bytes - issued * STAGE_PIECE);
            if (c->stage_busy[issued % STAGE_R]) {  // (an upload before this call may still be reading the buffer)
                err = hipEventSynchronize(c->ev_stage[issued % STAGE_R]);
                c->stage_busy[issued % STAGE_R] = false;
                if (err != hipSuccess) break;
            }
            err = hipMemcpyAsync(c->h_stage + (issued % STAGE_R) * STAGE_PIECE, static_cast<const uint8_t *>(src_dev) + issued * STAGE_PIECE,
                                 len, hipMemcpyDeviceToHost, c->stream);
            if (err == hipSuccess) err = hipEventRecord(c->ev_stage[issued % STAGE_R], c->stream);
        }
        if (err == hipSuccess) err = hipEventSynchronize(c->ev_stage[i % STAGE_R]);  // (piece i has arrived; nothing is pending on its buffer)
        go.store((long long)i, std::memory_order_release);  // (on an error the workers still run through every piece, so that they end)
        copy_out(i, 0);
        while (done.load(std::memory_order_acquire) < (unsigned long long)T * (i + 1)) std::this_thread::yield();
    }
    for (std::thread &t : pool) t.join();
    if (err != hipSuccess) return fail(c, BPE_E_HIP, "pinned download: %s", hipGetErrorString(err));
    return BPE_OK;
}

int read_state(bpe_ctx *c, DevState *out) {
    HIPCHK(c, hipMemcpyAsync(out, c->d_st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BPE_OK;
}

}  // namespace
