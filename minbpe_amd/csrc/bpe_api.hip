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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <deque>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "bpe_hip.h"
#include "bpe_kernels.hip"

using namespace bpe;
using namespace bpe::bpe_g4;  // (host code names the 1024-id geometry unless it says otherwise: GK below)

#ifndef REPACK_DEN
#define REPACK_DEN 32  // re-pack the slots when their fill drops below (REPACK_DEN-1)/REPACK_DEN
#endif

namespace {
thread_local std::string g_create_err;

struct ProfEv {
    int kind;
    hipEvent_t e0, e1;
    uint64_t bytes;
    uint32_t weight;  // sampled iterations stand for this many
};
}  // namespace

// the parts, in order (one translation unit: the kernels above are launched from all of them)
#include "api/api_ctx.hip"
#include "api/api_basic.hip"
#include "api/api_train.hip"
#include "api/api_encode.hip"
#include "api/api_decode.hip"
#include "api/api_dp.hip"
#include "api/api_rccl.hip"
#include "api/api_prof.hip"
