// bpe_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the BPE hot path.
//
// Written for 64-wide wavefronts, 256 CUs in 8 XCDs, 160 KiB LDS per CU and an
// HBM3E-bound workload: every kernel here is integer / byte work whose roofline
// is HBM bandwidth, so the rules that matter are coalesced 16 B-per-lane
// accesses, LDS pre-aggregation of atomics, and keeping the id stream resident.
// No MFMA anywhere.
//
// Data layout
//   id stream : uint32 words.  bits 0..25 = token id, bits 26..30 = weight
//               exponent of the token's chunk (0 unless chunks were de-duplicated),
//               bit 31 = "this token starts a chunk" (regex.py:44: pairs never
//               span chunks).  A pair (p, p+1) exists iff word[p+1] has bit 31
//               clear, so "(next word & ~weight bits) == b" is a complete test.
//   pair table: dense row-major uint32 matrix count[a][b], stride = vcap.
//               288 GB of HBM makes the dense form affordable (vocab 32000 ->
//               4.1 GB) and turns per-iteration table maintenance into four
//               dense vectors (delta mode) -- see k_apply_delta.
//   row maxima: rowmax[a] = max_b count[a][b]; argmax scans V values, not V^2.
//
// Reference semantics restated here:
//   F1 get_stats counts every adjacent pair        -> k_pair_count_*
//   F2 merge is greedy left-to-right               -> mbit() / run-parity scan
//   F3 argmax ties go to the earliest first occurrence -> k_select
//
// The kernels live in kernels/*.hip, included below in dependency order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bpe_device.h"

#include "kernels/k_common.hip"
#include "kernels/k_load.hip"
#include "kernels/k_stats.hip"
#include "kernels/k_table.hip"
// the parts that know the slot geometry, once per geometry (bpe_device.h: BPE_GEOMETRY)
#define BPE_G bpe_g4
#include "kernels/k_index.hip"
#include "kernels/k_select.hip"
#include "kernels/k_merge.hip"
#include "kernels/k_lookback.hip"
#include "kernels/k_slots.hip"
#include "kernels/k_slots2.hip"
#include "kernels/k_lean.hip"
#include "kernels/k_chain.hip"
#include "kernels/k_pool.hip"
#include "kernels/k_step.hip"
#include "kernels/k_dp.hip"
#undef BPE_G
#define BPE_G bpe_g1
#include "kernels/k_index.hip"
#include "kernels/k_select.hip"
#include "kernels/k_merge.hip"
#include "kernels/k_lookback.hip"
#include "kernels/k_slots.hip"
#include "kernels/k_slots2.hip"
#include "kernels/k_lean.hip"
#include "kernels/k_chain.hip"
#include "kernels/k_pool.hip"
#include "kernels/k_step.hip"
#include "kernels/k_dp.hip"
#undef BPE_G
#include "kernels/k_encode.hip"
#include "kernels/k_decode.hip"
#include "kernels/k_util.hip"
