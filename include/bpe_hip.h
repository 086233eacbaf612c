/*
 * bpe_hip.h -- C-ABI of libbpe_hip.so, the MI355X (gfx950) BPE train/encode engine.
 *
 * This is the drop-in boundary for minbpe's hot path.  The reference
 * (karpathy/minbpe) has no FFI of its own: its boundary is the Python surface
 * minbpe/__init__.py:1-4.  Each entry point below names the reference code it
 * replaces; minbpe_amd/ (Python, ctypes) mirrors the reference classes on top.
 *
 * Conventions
 *   - every call returns BPE_OK (0) or a negative status; no C++ exception
 *     crosses the ABI; bpe_last_error() gives a human-readable message.
 *   - the caller owns every host buffer for the duration of the call; the
 *     library owns all device memory, tied to the ctx, freed by bpe_destroy.
 *   - a ctx is bound to one GPU and is not thread-safe.  One ctx per GPU.
 *   - token ids are int32 (minbpe: Python ints; new id = 256 + i, basic.py:37).
 *   - chunk_offsets: n_chunks START offsets into the byte/id stream, ascending,
 *     chunk_offsets[0] == 0; NULL (n_chunks ignored) = one chunk.  Pairs never
 *     span chunks (regex.py:44,60).
 */
#ifndef BPE_HIP_H
#define BPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPE_OK 0
#define BPE_E_HIP (-1)         /* a HIP runtime call failed (see bpe_last_error) */
#define BPE_E_ARG (-2)         /* bad argument */
#define BPE_E_EMPTY_STATS (-3) /* stats dict empty: minbpe raises ValueError from max() (basic.py:35, regex.py:56) */
#define BPE_E_STATE (-4)       /* call out of order (e.g. train before load) */
#define BPE_E_CAP (-5)         /* caller's output buffer too small */
#define BPE_E_LIMIT (-6)       /* size beyond what this build supports */
#define BPE_E_INTERNAL (-7)    /* device-side consistency check failed */

typedef struct bpe_ctx bpe_ctx;

/* ---- lifetime -------------------------------------------------------------- */
int bpe_create(int device_id, bpe_ctx **out);
void bpe_destroy(bpe_ctx *ctx);
const char *bpe_last_error(bpe_ctx *ctx); /* ctx may be NULL: last create error */
/* Run all subsequent work of this ctx on an existing hipStream_t (e.g. torch's
 * current stream, so collectives issued by the host order with our kernels). */
int bpe_set_stream(bpe_ctx *ctx, void *hip_stream);
/* Knobs: "mode" = 0 recount (get_stats every iteration, the literal reference
 * loop) | 1 delta (pair table kept current by the merge pass).  "profile" =
 * 1 records hipEvents around every hot kernel.  Unknown names -> BPE_E_ARG.
 * Variants of the training loop, all producing identical merges (kept for
 * cross-checks and measurement; the defaults are the fast ones):
 *   "slots"  0 contiguous stream | 1 4096-id slots | 2 one-wave slots, in place (default)
 *   "sparse" 0 every pass visits every slot | 1 auto (default) | 2 always through the index
 *   "lean"   0 every iteration takes the general five-launch path | 1 three-launch lean iterations once a
 *            pair's count is <= "lean_count" (2^20; default) | 2 from the first merge on (tests)
 *   "lean_sum" (1: a lean selection works from the previous table update's per-wave records),
 *   "lean_chain" (1: tied pairs are merged off the list one selection made), "aa_sparse" (1: an a == b
 *   pass visits the slots the index names and keeps the index current), "enc_cache" (1), "enc_chain" (1),
 *   "sparse_ratio" (1), "tie_index" (1), "rep_min" (4), "rep_max" (8: log2 of delta replicas),
 *   "lds_delta" (1), "depth" (8: iterations the host runs ahead), "prof_stride" (64), "merge", "k1",
 *   "lb_tune", "scan_sup" (1024: the three-pass merge scans the tile summaries of streams of more tiles
 *   than this in three small launches instead of one workgroup). */
int bpe_set_option(bpe_ctx *ctx, const char *name, int64_t value);

/* ---- input ----------------------------------------------------------------- */
/* text.encode("utf-8") -> list(bytes)  (basic.py:25-26; per chunk regex.py:44).
 * Uploads the bytes; they stay resident so bpe_train can be re-run. */
int bpe_load_bytes(bpe_ctx *ctx, const uint8_t *bytes, uint64_t n,
                   const uint64_t *chunk_offsets, uint64_t n_chunks);
/* The same with a weight per chunk: every pair inside chunk c counts
 * 2^weight_exp[c] times (weight_exp[c] <= 31).  A chunk that occurs w times in the text
 * (regex.py:41-44 keeps all w copies) can be loaded once per set bit of w -- see
 * bpe_dedup_chunks, which builds exactly that list in first-appearance order, so that
 * counts AND the first-occurrence tie-break are those of the full list.  Stream lengths
 * reported by bpe_train then refer to the de-duplicated stream. */
int bpe_load_bytes_weighted(bpe_ctx *ctx, const uint8_t *bytes, uint64_t n,
                            const uint64_t *chunk_offsets, uint64_t n_chunks,
                            const uint8_t *weight_exp);
/* Arbitrary id lists, for the module-level get_stats()/merge() drop-ins
 * (base.py:13-41 called on user lists). */
int bpe_load_ids(bpe_ctx *ctx, const int32_t *ids, uint64_t n,
                 const uint64_t *chunk_offsets, uint64_t n_chunks);

/* ---- single-step hot functions (parity tests, module-level drop-ins) ------- */
/* get_stats(ids) over all chunks into one table (base.py:13-22, regex.py:51-54).
 * Also records each pair's first position (dict insertion order, F3). */
int bpe_get_stats(bpe_ctx *ctx, uint64_t *n_pairs_out);
/* Read back the table of the last bpe_get_stats, unordered; sort by first_pos
 * to obtain the reference's dict order. */
int bpe_read_stats(bpe_ctx *ctx, int32_t *a, int32_t *b, uint64_t *cnt,
                   uint64_t *first_pos, uint64_t cap, uint64_t *n_out);
/* max(stats, key=stats.get) with first-occurrence tie-break (basic.py:35) on
 * the current ids: returns the pair and its count, BPE_E_EMPTY_STATS if none. */
int bpe_argmax(bpe_ctx *ctx, int32_t *a, int32_t *b, uint64_t *count);
/* merge(ids, pair, idx) on every chunk (base.py:25-41, regex.py:60). */
int bpe_merge(bpe_ctx *ctx, int32_t a, int32_t b, int32_t idx, uint64_t *new_len);
int bpe_len(bpe_ctx *ctx, uint64_t *n);
/* Current ids, start flags stripped. */
int bpe_read_ids(bpe_ctx *ctx, int32_t *out, uint64_t cap);
/* Current chunk start offsets (positions in the current id stream). */
int bpe_read_chunk_starts(bpe_ctx *ctx, uint64_t *out, uint64_t cap, uint64_t *n_out);

/* ---- the training loop ----------------------------------------------------- */
/* BasicTokenizer.train / RegexTokenizer.train loop (basic.py:31-45,
 * regex.py:49-66) on the loaded bytes, entirely on device.
 *   pairs_out[2*i], pairs_out[2*i+1] : pair merged at iteration i (idx 256+i)
 *   counts_out[i]                    : stats[pair] (the verbose print, F10)
 *   len_out[i]                       : total ids after merge i
 *   iter_ms_out                      : optional (NULL ok) per-iteration device ms
 *   n_done                           : merges completed
 * Returns BPE_E_EMPTY_STATS when the pair table empties before num_merges
 * (n_done tells where); the Python layer raises ValueError like the reference. */
int bpe_train(bpe_ctx *ctx, int32_t num_merges, int32_t *pairs_out,
              uint64_t *counts_out, double *iter_ms_out, uint64_t *len_out,
              int32_t *n_done);

/* ---- data-parallel training over sharded chunks ------------------- */
/* One ctx per rank; each rank bpe_load_bytes() its contiguous range of chunks.
 * Every rank keeps a replica of the GLOBAL pair table; per iteration the host
 * all-reduces two tiny device buffers (RCCL through torch.distributed, see
 * minbpe_amd/dist.py) -- ids and the table never move:
 *
 *   bpe_dp_begin(num_merges, rank, nranks)   widen + local byte-pair counts
 *   [all-reduce SUM   table   (int32 x 2 x 65536: every count as two 16-bit limbs, low halves then high halves,
 *                              so that the sum over up to 1024 ranks cannot wrap)]
 *   bpe_dp_table_ready()                     the GLOBAL counts from the summed limbs; BPE_E_LIMIT on EVERY rank if a
 *                                            pair occurs 2^32 times or more in the whole job (counts are 32-bit)
 *   for i in range(num_merges):
 *       bpe_dp_select(i)                     arg-max on the replica; local tie-break candidate
 *       [all-reduce MIN   tiekey  (int64 x 3)]   lowest (rank, position) wins the tie (F3/F5);
 *                                               word 2 = -(device status): any rank's failure stops all
 *                                               (a status raised inside merge i travels with exchange i + 1:
 *                                               the peers stop one merge after the failing rank)
 *       bpe_dp_merge(i)                      merge locally, produce the 4 delta vectors
 *       [all-reduce SUM   delta   (int32 x delta_count)]
 *       bpe_dp_apply(i)                      fold them into the replica
 *   bpe_dp_poll(i, ...) any time after bpe_dp_merge(i): the iteration's record
 *   bpe_dp_end()
 * All calls only enqueue work on the ctx's stream (bpe_set_stream) except
 * bpe_dp_poll, which waits for the device to report iteration i.
 * EVERY rank must issue this same schedule for every i in [0, num_merges), also after one of
 * its own polls reported a failure: the remaining iterations are no-ops on the device, and a
 * rank that stopped issuing collectives would leave its peers blocked in theirs. */
int bpe_dp_begin(bpe_ctx *ctx, int32_t num_merges, int32_t rank, int32_t nranks);
int bpe_dp_buffers(bpe_ctx *ctx, void **table, uint64_t *table_count, void **delta,
                   uint64_t *delta_count, void **tiekey);
int bpe_dp_table_ready(bpe_ctx *ctx);
int bpe_dp_select(bpe_ctx *ctx, int32_t iter);
int bpe_dp_merge(bpe_ctx *ctx, int32_t iter);
int bpe_dp_apply(bpe_ctx *ctx, int32_t iter);
int bpe_dp_poll(bpe_ctx *ctx, int32_t iter, int32_t *a, int32_t *b, uint64_t *count,
                uint64_t *local_len, int32_t *status);
int bpe_dp_end(bpe_ctx *ctx);
/* The same loop driven from inside the library with RCCL called directly (librccl is
 * dlopen'ed): rank 0 makes a 128-byte id (bpe_comm_unique_id), the application
 * broadcasts it by whatever means it has, every rank calls bpe_comm_init, then
 * bpe_dp_train.  Identical results on every rank; len_out holds GLOBAL lengths. */
/* 1 if librccl could be loaded in this process (ranks should agree on this, e.g. with a MIN
 * all-reduce over their bootstrap transport, BEFORE any of them enters bpe_comm_init: a rank
 * without the library would leave the others blocked in ncclCommInitRank). */
int bpe_comm_available(void);
int bpe_comm_unique_id(uint8_t *out128);
int bpe_comm_init(bpe_ctx *ctx, int32_t rank, int32_t nranks, const uint8_t *id128);
int bpe_comm_destroy(bpe_ctx *ctx);
int bpe_dp_train(bpe_ctx *ctx, int32_t num_merges, int32_t *pairs_out, uint64_t *counts_out,
                 uint64_t *len_out, int32_t *n_done);
/* bpe_dp_train runs the single-GPU engine's CHAIN STEPS across the ranks: a step merges the next 1..K pairs
 * the reference would merge (the tied pairs at the maximum in order of first occurrence, as long as they share no
 * token), and costs two all-reduces whatever K is:
 *   MIN  int64 x 98              a tie's first occurrences, rank << 40 | local position per tied pair (word 0 = -status,
 *                                word 1 = -1 from a rank that cannot order its share: the general path decides instead)
 *   SUM  int32 x (2 K_cap S + 64)   the batch's delta vectors (S = ids in use, rounded to 64) + per-pair adj words +
 *                                one status word: a failure inside rank r's merge pass is known to every rank BEFORE
 *                                the table update of that same step -- all ranks stop at the same merge
 * (the merges before the tied regime, and every a == b merge, take the per-merge schedule above).
 * bpe_dp_train_cb is the same loop with the collectives handed to the caller: fn(user, device_buffer, count, dtype,
 * op, hip_stream) must all-reduce `count` elements IN PLACE across the ranks, ordered after the work already enqueued on
 * hip_stream and before whatever is enqueued next (enqueue it on that stream, or synchronise), and return 0 -- how
 * torch.distributed (or a test's host-side reduction) drives the loop without librccl in the library.  Every rank
 * calls fn the same number of times with the same counts, also after a rank-local failure. */
#define BPE_DT_INT32 0
#define BPE_DT_INT64 1
#define BPE_OP_SUM 0
#define BPE_OP_MIN 1
typedef int (*bpe_allreduce_fn)(void *user, void *device_buffer, uint64_t count, int32_t dtype, int32_t op,
                                void *hip_stream);
int bpe_dp_train_cb(bpe_ctx *ctx, int32_t num_merges, int32_t rank, int32_t nranks, bpe_allreduce_fn fn, void *user,
                    int32_t *pairs_out, uint64_t *counts_out, uint64_t *len_out, int32_t *n_done);

/* ---- encode ----------------------------------------------------------------- */
/* _encode_chunk for a batch of chunks (regex.py:92-121; basic.py:57-74 when
 * n_chunks == 1).  merges: 2*M int32, in PRIORITY order (the reference's
 * `min(stats, key=merges.get)` picks the lowest merges[pair] value; pass the
 * pairs sorted by that value); the pair at position r merges to id merge_ids[r]
 * (NULL: 256 + r).  ids_out needs n entries; out_offsets n_chunks+1 entries
 * (token offset of each chunk in ids_out).  Invalidates the ids loaded by
 * bpe_load_bytes/bpe_load_ids (buffers are shared). */
int bpe_encode_batch(bpe_ctx *ctx, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                     const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets,
                     uint64_t n_chunks, int32_t *ids_out, uint64_t *out_offsets,
                     uint64_t *n_out);
/* The same with every large buffer already ON THE DEVICE (bytes, chunk_offsets, ids_out with room for n ids, out_offsets
 * with n_chunks + 1 entries are device pointers; merges / merge_ids stay host arrays): the kernels read the caller's
 * offsets and write the caller's outputs in place; only the rank table and a few counters cross PCIe.  For callers
 * that keep their batches in HBM (a data loader on the GPU, torch tensors: tensor.data_ptr()).  Offsets are validated
 * on the device (ascending, <= n).  Synchronises the ctx's stream before returning. */
int bpe_encode_batch_resident(bpe_ctx *ctx, const int32_t *merges, const int32_t *merge_ids, int32_t M,
                              const uint8_t *d_bytes, uint64_t n, const uint64_t *d_chunk_offsets,
                              uint64_t n_chunks, int32_t *d_ids_out, uint64_t *d_out_offsets,
                              uint64_t *n_out);
/* Whether bpe_encode_batch's encoder WITHOUT the chunk cache (option enc_cache = 0) keeps tokens and ranks
 * in 16-bit columns for this merge table (host logic, no GPU): every rank below 65535 and every token id
 * -- merge_ids[r], or 256 + r when merge_ids is NULL -- below 65536.  The default (cached) encoder is
 * 32-bit throughout. */
int bpe_encode_uses_16bit(const int32_t *merge_ids, int32_t M);

/* ---- decode -------------------------------------------------------- */
/* `b"".join(vocab[idx] for idx in ids)` (basic.py:51-55, regex.py:78-90, gpt4.py:87-92)
 * for a batch of token ids.  The vocab is a table resident in HBM: entry i holds the
 * bytes vocab_bytes[vocab_offsets[i] .. vocab_offsets[i+1]), i in [0, V).  Sparse ids
 * (special tokens) are the caller's to map onto table indices.
 *   bpe_decode_set_vocab   upload the table (kept until replaced)
 *   bpe_decode_batch       ids -> bytes, left resident; *n_bytes = their number.  An id
 *                          outside [0, V) fails with BPE_E_ARG and *bad_index = the first
 *                          such position (the reference raises KeyError / ValueError there)
 *   bpe_decode_read        copy the bytes out; with doc_token_offsets (k token positions,
 *                          each <= n) also the byte offset of each of those positions. */
int bpe_decode_set_vocab(bpe_ctx *ctx, const uint8_t *vocab_bytes, const uint64_t *vocab_offsets, int32_t V);
int bpe_decode_batch(bpe_ctx *ctx, const int32_t *ids, uint64_t n, uint64_t *n_bytes, uint64_t *bad_index);
int bpe_decode_read(bpe_ctx *ctx, uint8_t *out, uint64_t cap, const uint64_t *doc_token_offsets,
                    uint64_t k, uint64_t *doc_byte_offsets_out);

/* ---- measurement ------------------------------------------------------------ */
#define BPE_PROF_WIDEN 0
#define BPE_PROF_PAIR_COUNT 1
#define BPE_PROF_ARGMAX 2   /* rowmax + argmax + tie-break + finalize */
#define BPE_PROF_MERGE 3    /* merge pass(es) */
#define BPE_PROF_TABLE 4    /* delta apply / table clear */
#define BPE_PROF_ENCODE 5
#define BPE_PROF_DECODE 6
#define BPE_PROF_NKINDS 7
/* Accumulated since the last bpe_prof_reset: device ms (hipEvents on the ctx's
 * stream), launches, algorithmic bytes (4 B per id read or written,
 * tables and flags not counted). */
int bpe_prof_reset(bpe_ctx *ctx);
int bpe_prof_read(bpe_ctx *ctx, double *ms, uint64_t *launches, uint64_t *alg_bytes);
/* How the last bpe_train ran its merge passes: out[0] = dense passes (every slot of the stream
 * is visited), out[1] = sparse passes (only the slots the inverted slot index cannot rule out),
 * out[2] = builds of that index, out[3] = slots of the stream at the end. */
int bpe_train_stats(bpe_ctx *ctx, uint64_t *out4);
/* The same, extended: out[4] = merges done by lean iterations or chain steps (three launches per merge or per
 * batch of merges, the pair table updated at the merge sites themselves; options "lean", "chain"), out[5] =
 * merges they handed back to the general path (pairs with a == b, ties they could not settle), out[6] = merges
 * that needed no selection of their own (their pair came off the list of tied pairs an earlier selection made:
 * the reference merges those in order of first occurrence while their counts stand), out[7] = chain steps,
 * out[8] = chain steps that selected (gathered the pool of pairs anew; the others took their pairs off it), out[9] = ids
 * per slot the stream ended in (1024; 256 once it was re-packed for sparse passes, option "small_slots"), out[10] =
 * chain steps that were ONE launch (selection, merge pass and table update as phases of one resident grid: option
 * "fuse_step").  Writes min(n, 11) values. */
int bpe_train_stats_ex(bpe_ctx *ctx, uint64_t *out, int n);

/* ---- text.encode("utf-8") by all host threads (host, no GPU needed) ------------------ */
/* The first statement of every train() / encode() of the reference (basic.py:25, regex.py:44): an array of code
 * points of `kind` = 1, 2 or 4 bytes each (what a CPython str holds) as UTF-8, counted and written by segments in
 * parallel.  out may be NULL to only count (*n_bytes); BPE_E_CAP if cap is too small; BPE_E_ARG on a lone surrogate
 * or a value above 0x10FFFF (str.encode raises there: the caller leaves such a text to it).  threads <= 0: all cores. */
int bpe_utf8_encode(int kind, const void *code_points, uint64_t n_chars, uint8_t *out, uint64_t cap,
                    uint64_t *n_bytes, int threads);

/* ---- native pre-split (host, no GPU needed) ----------------------------- */
/* regex.findall(pattern, text) for the two GPT split patterns (regex.py:18-19, 41, 114),
 * as chunk START byte offsets into the UTF-8 text.  which: 2 = GPT-2 pattern, 4 = GPT-4
 * pattern (any other pattern stays with the `regex` module).  Both patterns match every
 * character, so the chunks partition the text.  starts_out may be NULL to only count;
 * returns BPE_E_CAP (with *n_chunks set) if cap is too small.  threads <= 0: all cores. */
int bpe_split(int which, const uint8_t *utf8, uint64_t n, uint64_t *starts_out, uint64_t cap,
              uint64_t *n_chunks, int threads);

/* The same for a batch of documents laid out back to back (document d starts at doc_offsets[d];
 * the last one ends at n): every document is split on its own, as a loop of encode() calls over
 * the documents would (regex.py:111-121).  doc_first_chunk (n_docs + 1 entries, may be NULL)
 * receives the index of each document's first chunk in starts_out. */
int bpe_split_docs(int which, const uint8_t *utf8, uint64_t n, const uint64_t *doc_offsets,
                   uint64_t n_docs, uint64_t *starts_out, uint64_t cap, uint64_t *n_chunks,
                   uint64_t *doc_first_chunk, int threads);

/* ---- chunk de-duplication (host, no GPU needed) ---------------------------- */
/* In: the chunk list of regex.py:41-44 as bytes + chunk START offsets (chunk c ends where
 * chunk c+1 starts, the last one at n).  Out: the distinct chunks in order of first
 * appearance, each emitted once per set bit k of its multiplicity with weight_exp = k (so
 * the emitted weights 2^k sum to the multiplicity).  The output never exceeds the input:
 * out_bytes needs n bytes, out_offsets and out_weight_exp need n_chunks entries.
 * threads <= 0: all cores. */
int bpe_dedup_chunks(const uint8_t *bytes, uint64_t n, const uint64_t *chunk_offsets, uint64_t n_chunks,
                     uint8_t *out_bytes, uint64_t *out_offsets, uint8_t *out_weight_exp,
                     uint64_t *n_out_bytes, uint64_t *n_out_chunks, uint64_t *n_distinct, int threads);

/* ---- host utilities (no GPU needed) ------------------------------------------ */
/* Deterministic synthetic UTF-8 text (SURVEY 8d synth_text): explicit
 * splitmix64, integer tables only.  Writes exactly n bytes, valid UTF-8. */
int bpe_synth_text(uint8_t *out, uint64_t n, uint64_t seed);
/* Library version / build info string. */
const char *bpe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BPE_HIP_H */
