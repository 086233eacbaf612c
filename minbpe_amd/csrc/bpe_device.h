// bpe_device.h -- constants and device-visible structs shared by the kernels
// (bpe_kernels.hip) and the host side of the C-ABI (bpe_api.hip).
#pragma once
#include <stdint.h>

namespace bpe {

// id-stream word format
constexpr uint32_t FLAG = 0x80000000u;          // bit 31: token starts a chunk
constexpr uint32_t IDMASK = 0x03FFFFFFu;        // bits 0..25: token id
// bits 26..30: weight exponent e of the token's chunk -- every pair inside the chunk counts
// 2^e times (chunk de-duplication; 0 everywhere for an unweighted stream)
constexpr int WSHIFT = 26;
constexpr uint32_t WMASK = 0x7C000000u;
constexpr uint32_t NWMASK = ~WMASK;             // flag + id: what a pair comparison looks at
constexpr uint32_t INVALID_WORD = 0xFFFFFFFFu;  // positions >= n inside a tile
constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;     // LDS cache: free slot
constexpr unsigned long long NOPOS = ~0ull;

// merge tile geometry: 4 waves x MJ stripes x 64 lanes x 4 ids.  The kernels that know the geometry (kernels/k_index ..
// k_dp, bpe_kernels.hip) are compiled ONCE PER GEOMETRY, each in its own namespace: bpe_g4 -- MJ = 4, slots of 1024 ids:
// the dense sweeps of the early merges want 4 KiB per wave in flight -- and bpe_g1 -- MJ = 1, slots of 256 ids: once
// passes are sparse a merge site costs the slot it sits in, and a slot of 1 KiB is a quarter of the traffic and of the
// per-slot work.  The host re-packs the stream from one into the other when the inverted index is first built
// (option "small_slots") and launches the kernels of the geometry the stream is in (api_ctx.hip: GK).
constexpr int MT = 256;
#define BPE_GEOMETRY(NS, MJV, IDXH_LOG2)                                                                         \
    namespace NS {                                                                                               \
    constexpr int MJ = MJV; /* <= 8: a group of 64 tiles must fit 20-bit aggregates */                           \
    constexpr int WAVE_SPAN = MJ * 256;                                                                          \
    constexpr int TILE = (MT / 64) * WAVE_SPAN; /* first slotted form / contiguous kernels: 4 waves per tile */ \
    /* Slot size of the second slotted form: what ONE WAVE holds (MJ stripes x 64 lanes x 4 ids).  A slot is   \
       read, compacted and written back by a single wave -- no barrier anywhere in the pass */                  \
    constexpr int TILE2 = WAVE_SPAN;                                                                             \
    /* inverted slot index (k_index.hip): buckets of a slot group's pair filter */                              \
    constexpr uint32_t IDX_H = 1u << IDXH_LOG2;                                                                  \
    constexpr int IDX_SHIFT = 32 - IDXH_LOG2;                                                                    \
    }
BPE_GEOMETRY(bpe_g4, 4, 15)
#ifndef BPE_G1_IDXH_LOG2
#define BPE_G1_IDXH_LOG2 13
#endif
BPE_GEOMETRY(bpe_g1, 1, BPE_G1_IDXH_LOG2)
constexpr int TILE2_MAX = 1024, TILE2_MIN = 256;

// pair-count kernels: one 1024-thread workgroup per CU using all 128 KiB of dynamic LDS
constexpr int PC_BITS = 14;       // general kernel: 2^14 {key,count} slots
constexpr int PC_THREADS = 1024;
constexpr int PC_LDS_BYTES = 131072;
constexpr int PCB_ROUND = 61440;  // byte-stream kernel: positions between flushes (< 65536)

constexpr int TIE_CAP = 96;        // tied pairs carried explicitly; more -> table lookup
constexpr int ARGMAX_ROWS = 2048;  // rows attaining the max examined by k_select
constexpr uint32_t TIE_WINDOW0 = 0;  // positions k_select's block 0 searches alone on a tie (0: all blocks sweep together)
constexpr int TIE_BLOCKS = 256;     // k_select blocks: block 0 decides, all of them sweep the stream on a tie
constexpr int ROW_BLOCKS = 64;      // extra k_apply_delta blocks that recompute queued row maxima
constexpr int DELTA_REPL = 256;    // replica blocks of the delta vectors (spreads hot atomics; a chain step's batch
                                   // gives each of its pairs CH_RSTRIDE of them, k_chain.hip)
// Device-scope atomics execute at the memory channel that owns the address, ~11 ns apiece and ONE AT
// A TIME per channel (measured: a pass whose atomics fall on few channels runs at a fraction of the
// ~10 G atomics/s the whole chip sustains).  Replica r of the delta vectors starts at
// r * (4 * stride + DELTA_SKEW) words: the skew of one 256-byte line walks the replicas of one hot
// token over all the channels instead of the few a power-of-two stride lands on; the removal
// counters of a merge pass sit one per 256-byte line for the same reason.
constexpr int DELTA_SKEW = 64;
constexpr int REMOVED_STRIDE = 64;  // words between the 256 removal counters
// While every id is below LDSD_CAP (the first ~1700 merges, where passes have millions of sites)
// the a != b kernels add their delta into LDS tables and flush them once per workgroup.
constexpr int LDSD_CAP = 1920;
__host__ __device__ inline size_t delta_rep_off(uint32_t r, uint32_t stride) {
    return (size_t)r * (4 * (size_t)stride + DELTA_SKEW);
}

// Chain steps (k_chain.hip): the tied pairs at the maximum are kept as a LIST in first-occurrence order, and the
// longest prefix of it whose pairs have a != b and share no token -- at most CH_KMAX of them -- is merged in ONE pass.
constexpr int CH_KMAX = 16;                    // (array sizes)
constexpr int CH_KSWEEP = 15;                  // most pairs one sweep merges: a word's pair number + 1 is a nibble (merge_chain_wave)
constexpr int CH_RSTRIDE = 16;                 // replica blocks set aside per pair of a batch: all of them while its pairs have
constexpr uint32_t CH_REP_COUNT = 16384;       // more sites than this (hot tokens queue ~11 ns per same-address atomic) ...
constexpr int CH_REP = 4;                      // ... this many otherwise (the table update folds every replica it is told to)
constexpr int CH_RMV = 16;                     // removal counters per pair of a batch (of the 256)
constexpr int DP_KCAP_MAX = CH_KSWEEP;         // sharded chain steps: most pairs of a batch (option dp_kcap; the SUM payload grows with it: 2 K S words)
constexpr int DP_KCAP_DEFAULT = 8;
static_assert(CH_KMAX * CH_RSTRIDE <= DELTA_REPL, "a batch's delta vectors must fit the replica blocks");
static_assert(CH_KMAX * CH_RMV <= 256, "removal counters of a batch");
static_assert(CH_KSWEEP < 16 && CH_KSWEEP <= CH_KMAX, "pair number + 1 must fit a nibble");
constexpr uint32_t CH_FULL = 0, CH_LIST = 1;   // DevState::sel_mode
constexpr int DP_KEY_WORDS = TIE_CAP + 2;      // sharded chain steps: int64 words of the MIN all-reduce (k_pool_sel)

// encode: one chunk per lane, token lists in lane-private LDS columns
constexpr int ENC_THREADS = 256;
constexpr int ENC_LMAX = 32;  // longer chunks take the stream-wide path
constexpr int SCAN_TILE = 4096;

constexpr uint32_t ST_OK = 0, ST_EMPTY = 1, ST_INTERNAL = 2, ST_LOOKBACK = 3;
constexpr uint32_t ST_DEFER = 4;  // IterRec only: a lean iteration handed the merge back to the general path
constexpr uint32_t EPOCH_MASK = 0xFFFFFu;  // look-back descriptors carry a 20-bit launch tag
constexpr uint32_t LOOKBACK_SPINS = 1u << 20;  // bounded wait for a predecessor tile

// A view of the id stream.  meta == nullptr: contiguous, b0[0..n).  Otherwise
// slotted: slot t occupies [t*TILE, t*TILE + len) of buffer b0 or b1, with
// meta[t] = len | (which buffer) << 31.  Stream order = slot order, so a
// slot-space position p = t*TILE + offset still orders first occurrences.
struct SlotRef {
    const uint32_t *b0, *b1;
    const uint32_t *meta;
    unsigned long long T;
};

// Slotted stream, second form (the training loop's default, k_slots2.hip): one 32-byte header
// per slot instead of a meta word + a 16-byte header, so that a workgroup gets everything it
// needs from its neighbours in one coalesced 96-byte load.
struct SlotHdr {
    uint32_t w0, w1, w2;  // first three words of the slot (INVALID_WORD beyond its length)
    uint32_t meta;        // len | buffer << 31
    uint32_t l0, l1;      // second-to-last and last word (INVALID_WORD if len < 2 / len < 1)
    uint32_t pad0, pad1;
};
struct SlotRefH {
    const uint32_t *b0, *b1;
    const SlotHdr *hdr;
    unsigned long long T;
};
// a header waiting to be committed (sparse merge passes visit few slots and must not touch the
// header array their neighbours are reading)
struct StageRec {
    uint32_t t, pad[3];
    uint32_t h[8];
};
constexpr uint32_t ROWARG_MULTI = 0xFFFFFFFFu;  // rowarg[x]: several columns attain rowmax[x]

// one per ctx, in device memory
struct DevState {
    unsigned long long n[2];      // length of the id stream in ping-pong buffer 0/1
    unsigned long long firstpos;  // tie-break: earliest position holding a tied pair
    int32_t a, b;                 // pair being merged
    uint32_t count;               // its count
    uint32_t ntied;               // pairs tied at the max (TIE_CAP+1: too many to list)
    uint32_t found;               // (a,b) decided
    uint32_t status;              // ST_*
    int32_t tied[2 * TIE_CAP];
    int32_t fin_a, fin_b;         // the pair as finalised by the merge pass (read by k_apply_delta)
    unsigned long long removed;   // slotted merge: ids removed by the current pass
    unsigned long long apply_done;  // k_apply_delta blocks finished (monotonic; row blocks wait on it)
    uint32_t sel_tie;             // ... and whether a tie is open (every block then sweeps and takes a ticket)
    uint32_t sel_done;            // tickets of the sweeping blocks: the last one finalises the pair
    uint32_t adj;                 // delta format B: sites whose right neighbour starts another site
    uint32_t ncand;               // slots in the candidate list of this iteration's sparse pass (k_select)
    uint32_t gap;                 // some slot other than the last holds < 3 ids: sparse passes visit every slot
    uint32_t tlive;               // slots [tlive, T) were empty when the stream was last re-packed, and stay so: the host
                                  // sizes T from a length that is `depth` merges old, and nobody should walk that tail
    uint32_t defer;               // lean iterations (k_lean.hip): the decided pair needs the general path (a == b): this
                                  // iteration and everything enqueued behind it do nothing until the host has re-run it there
    // lean iterations: the pair k_apply_lean folded last and its new token -- rows scan_a, scan_b, scan_z are
    // re-scanned (and (scan_a, scan_b) retired) by the NEXT launch that needs the row maxima; 0xFFFFFFFF = none.
    // Re-scanning them again is harmless: a merged pair never re-forms.
    uint32_t scan_a, scan_b, scan_z;
    // lean iterations, chained merges (k_lean.hip): the pairs tied at the maximum when k_sel_lean last
    // decided, in the order of their first occurrences, up to the first one that shares a token with an
    // earlier one or has a == b.  chain[0] was that iteration's merge; the following iterations take
    // chain[chain_pos], chain_pos + 1, ... without selecting, as long as no merge of the chain created
    // a pair that reaches the tied count (chain_cut, set by k_apply_lean).
    uint32_t chain_n, chain_pos, chain_cut;
    uint32_t chain_taken;         // iterations that took their pair off the chain since the stream was loaded (statistics)
    uint32_t sel_ran;             // the last selection launch re-scanned every flagged row: the next lean merge pass clears the flags
    int32_t chain[2 * TIE_CAP];   // (chain steps: the tied-pair list itself, tl_n entries)
    // ---- chain steps (k_chain.hip): the device counts the merges itself, a step does 1..CH_KMAX of them ----
    uint32_t iter;                // merges done; the next new token is 256 + iter
    uint32_t num_merges;          // training stops there: later steps are no-ops
    uint32_t sel_mode;            // CH_FULL: the next step selects | CH_LIST: it takes its pairs off the list.  Written by
                                  // k_apply_chain / k_set_iter / k_clear_defer only, never by the launch that reads it
    uint32_t tl_n, tl_M, tl_skip; // the list: entries, their common count, how many at its head the last batch merged
    uint32_t bk, bz0;             // this step's batch: pairs, the new id of the first one
    int32_t ba[CH_KMAX], bb[CH_KMAX];
    uint32_t badj[CH_KMAX];       // delta format B, per pair of the batch: sites whose right neighbour starts a site of the SAME pair
    uint32_t bcnt[CH_KMAX];       // the pairs' counts (a batch may reach below the maximum)
    uint32_t brep;                // delta replicas per pair of this batch (CH_REP or CH_RSTRIDE)
    uint32_t bhm, bhm_key;        // the multiplier of the batch's first-token look-up table (merge_chain_wave), found once by the selection
                                  // instead of by every workgroup of the merge pass; valid iff bhm_key == bz0 << 8 | bk
    uint32_t dp_wait;             // sharded chain steps: the selection waits for the MIN all-reduce (k_pool_sel_dp finishes it)
    // the pool (k_pool.hip): every pair that counts pool_theta or more, pool_n entries in the ctx's pool buffer
    uint32_t pool_n, pool_theta;
    uint32_t pool_hint;           // the next selection launch may have to rebuild the pool: its row-scanning workgroups stay.
                                  // Every workgroup of a selection launch reads it at entry, so that launch never writes it:
    uint32_t pool_hint_next;      // ... the deciding workgroup leaves the next value here, the step's table update (or
                                  // k_set_iter / k_clear_defer_chain) makes it current -- a workgroup dispatched late must
                                  // not see a hint that differs from the one its peers acted on
    unsigned long long pool_epoch;  // launches that located entries so far (order keys of one epoch are comparable)
    // k_select: block 0 publishes its decision to the other blocks through this word.  They poll it
    // (hundreds of them): it sits alone in its 128-byte line, so that the polls do not queue up in
    // front of block 0's own accesses to the fields above.
    alignas(128) uint32_t sel_flag;
    uint32_t pad_flag_[31];
};

// ---- kernel argument blocks the host fills (the same types for both geometries) --------------------------------------
// the candidate list of a sparse pass / the index a tie is broken through (k_index.hip, k_select.hip)
struct CandArgs {
    const uint32_t *idx, *dirty;
    uint32_t *cand;
    uint32_t stride;     // of the index rows (groups allocated)
    uint32_t T;
    uint32_t enable;     // 0: this iteration's a != b pass is a dense one
    uint32_t tie_index;  // the index is live: block 0 of k_select breaks ties through it
    uint32_t tie_window; // block 0 first looks through the first TIE_WIN slots by itself (experiment)
    uint32_t aa;         // a list for a pair with a == b too (its pass then works through the list, k_merge_aa)
};
// an a != b merge pass of the second slotted form (k_slots2.hip, k_lean.hip, k_chain.hip)
struct AbArgs {
    uint32_t *b0, *b1;        // the two id buffers
    const SlotHdr *hdr_in;    // headers as they stand before this pass
    SlotHdr *hdr_out;         // dense: the other header array; sparse: unused
    StageRec *stage;          // sparse: staged headers, stage[t] for slot t ...
    uint32_t *smask;          // ... and [slot / 32]: which slots have one (no shared counter: every
                              // changed slot would queue up behind it, ~11 ns each)
    uint32_t T;
    DevState *st;
    uint32_t newid;
    uint32_t *delta;          // [replica][4][vcap]
    uint32_t vcap;            // row stride | log2(replicas) << 24
    uint32_t *idx;            // inverted index [IDX_H][istride] (word = slot / 32, bit = slot % 32), or nullptr
    uint32_t istride;
    const uint32_t *cand;     // sparse: the slots to visit (st->ncand of them, from k_select)
    uint32_t *removed;        // [256] ids removed by this pass, spread over counters (t & 255): one
                              // counter would serialise every changed slot of a dense pass (~11 ns each)
    uint32_t *dirty_n;        // reset here for the table update that follows
};
// the a == b pass (k_slots2.hip: k_merge_aa)
struct AaArgs {
    const uint32_t *b0, *b1;
    uint32_t *w0, *w1;
    const SlotHdr *hdr_in;
    SlotHdr *hdr_out;         // every slot's header is written here ...
    StageRec *stage;          // ... unless this is set (the a != b kernel of this iteration is the
                              // sparse one, whose staged headers get committed): changed slots only
    uint32_t *smask;
    uint32_t T;
    DevState *st;
    uint32_t newid;
    uint32_t *delta;
    uint32_t vcap;
    unsigned long long *sdesc;
    uint32_t epoch;
    uint32_t *dirty;          // index live and not kept current by this pass: [slot / 32], set for the slots it rewrites
    uint32_t *removed;        // [256]
    const uint32_t *cand;     // the slots to visit (st->ncand of them, from k_select: the slots whose filter admits
                              // (a,a), the slot before each, the marked ones), or nullptr: every slot
    uint32_t *idx;            // the index, kept current by this pass (the pairs it creates enter the filters), or nullptr
    uint32_t istride;
};
// an entry of the pool (k_pool.hip)
struct PoolEnt {
    uint32_t xy, c;            // x << 16 | y (training ids are below 65536), the pair's count
    unsigned long long key;    // epoch << 40 | first-occurrence position; 0 = no order known
};

constexpr uint32_t STEP_RING = 4096;  // step records form a ring (far more than the steps the host runs ahead: depth <= 64)
// one per chain step, written by the device into pinned host memory (the IterRecs of its merges are final before it)
struct StepRec {
    uint32_t first_iter, k;      // the merges this step did: first_iter .. first_iter + k - 1 (k == 0: none)
    uint32_t status, pad;        // ST_* (ST_DEFER: the general path must do merge first_iter); pad = the step's mode (CH_FULL / CH_LIST) | DevState::defer << 8
    unsigned long long new_len;
    unsigned long long seq;      // step + 1 once every field above is final
};

// one per training iteration, written by the device into pinned host memory
struct IterRec {
    int32_t a, b;
    uint32_t count;
    uint32_t status;
    unsigned long long new_len;
    unsigned long long seq;  // iteration + 1 once every field above is final
};

// a sparse chain step as one launch (k_step.hip): the arguments of its three phases -- selection (k_pool_sel's), merge
// pass (k_merge_chain's), table update (k_apply_chain's) -- and of the hand-overs between them
struct StepArgs {
    // selection
    uint32_t *rowmax, *mat;
    uint32_t stride;
    DevState *st;
    SlotRefH ref;
    CandArgs C;
    uint32_t *dbits;
    unsigned long long *res;   // row maxima on their way to the deciding workgroup
    uint32_t tag;              // this launch's tag (never repeats): validates res / req / pub words
    unsigned long long *req;   // a rebuild's request / answer words
    uint32_t kcap;
    PoolEnt *pool;
    uint32_t *gather;
    uint32_t hint_below;
    uint32_t nscan;            // workgroups 1 .. nscan do a rebuild's row work
    // merge pass
    AbArgs A;
    const uint32_t *idx_dirty;
    uint32_t use_index;
    uint32_t gm;               // workgroups 0 .. gm - 1 take part in the merge pass
    // table update
    uint32_t *delta;
    uint32_t dl;
    int par;
    IterRec *rec;
    StepRec *srec;
    uint32_t step;
    SlotHdr *hdr_cur;
    const StageRec *stage;
    uint32_t *removed;
    uint32_t *smask;
    uint32_t nwords;
    uint4 *sums;
    // hand-overs
    unsigned long long *pub;   // the published line (32 words: the batch itself)
    uint32_t *bar;             // grid-barrier counter: only ever grows ...
    uint32_t bar_target;       // ... to this value once every workgroup of this launch has arrived
    unsigned long long *stamps;  // nullptr, or [STEP_STAMP_RING][3 workgroups][16] clock stamps of the phases (BPE_STEP_STAMPS: where a step's time goes)
};
constexpr uint32_t STEP_STAMP_RING = 8192;

}  // namespace bpe
