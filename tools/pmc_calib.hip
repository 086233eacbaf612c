// pmc_calib.hip -- kernels with KNOWN byte counts in the access patterns the training kernels use, to calibrate
// the rocprofv3 HBM-traffic counters on gfx950 (profiles/README.md "counter calibration"):
//   FETCH_SIZE / WRITE_SIZE (derived; the guide's x2 holds for wide coalesced reads only) against
//   TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B / TCC_EA0_WRREQ_WRITE_ATOMIC_32B (raw, size-weighted:
//   a 64-byte request counts 2, a 128-byte one 4).
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib      (tools/gpu_calib.sh does it)
// Run under rocprofv3 --pmc <counters> --kernel-trace; prints one JSON line: kernel -> bytes requested per launch.
// Every pattern works on a 2 GiB buffer (eight times the 256 MiB Infinity Cache) and touches each byte at most once
// per launch, so "bytes requested" is also the least the memory system can move at the request granularity named.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x)                                                                      \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr size_t BUF = 2ull << 30;  // bytes
__device__ __forceinline__ uint32_t mix(uint32_t x) {  // a bijection of 32-bit values (odd multiply, xor-shift)
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}

// (1) wide coalesced streaming read: 16 B per lane, every byte of the buffer once
__global__ void cal_read16_stream(const uint4 *__restrict__ p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// (2) narrow coalesced streaming read: 4 B per lane (a wave reads 256 contiguous bytes per instruction)
__global__ void cal_read4_stream(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// (3) the lean / chain merge pass's reads: one WAVE per slot -- the slot's 4 KiB (four 16-byte loads per lane, 1 KiB per
// instruction) + three neighbouring 32-byte headers as six 16-byte pieces on lanes 0..5 (96 contiguous bytes) + one
// 4-byte mask word; slots picked at random (a bijection of the slot number), nslots of them.
__global__ void cal_read_slots(const uint32_t *__restrict__ words, const uint4 *__restrict__ hdr, const uint32_t *__restrict__ mask,
                               uint32_t nslots, uint32_t slot_mask, uint32_t *sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t s = wave; s < nslots; s += nw) {
        const uint32_t t = mix(s) & slot_mask;
        const uint32_t *src = words + (size_t)t * 1024;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + j * 256 + lane * 4);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
        if (lane < 6) {
            const uint4 h = hdr[2 * (size_t)t + lane];  // (its own array: 32 B per slot, three neighbours' worth)
            acc ^= h.x ^ h.w;
        }
        if (lane == 0) acc ^= mask[t];
    }
    if (acc == 0x12345678u) *sink = acc;
}
// (4) random 32-byte sectors: one lane reads one aligned 32-byte record (two 16-byte loads), all lanes of a wave at
// unrelated addresses (the staged header records, the encode table's entries)
__global__ void cal_read_sector32(const uint4 *__restrict__ p, uint32_t n, uint32_t rec_mask, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t r = mix(i) & rec_mask;
        const uint4 a = p[2 * r], b = p[2 * r + 1];
        acc ^= a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// (5) strided 4-byte gathers: lane l reads word [l * stride + c] -- a COLUMN of the pair table (k_apply_chain's
// (t, a) entries: one 4-byte word per 128 KB row)
__global__ void cal_read_column4(const uint32_t *__restrict__ p, uint32_t rows, uint32_t stride, uint32_t ncols, uint32_t *sink) {
    uint32_t acc = 0;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t c = 0; c < ncols; c++)
        if (tid < rows) acc ^= p[(size_t)tid * stride + (mix(c) % stride)];
    if (acc == 0x12345678u) *sink = acc;
}
// (6) wide coalesced streaming write: 16 B per lane
__global__ void cal_write16_stream(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// (7) random 32-byte record writes (two 16-byte stores by one lane): the header commits
__global__ void cal_write_sector32(uint4 *__restrict__ p, uint32_t n, uint32_t rec_mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t r = mix(i) & rec_mask;
        p[2 * r] = make_uint4(i, 0u, 0u, 0u);
        p[2 * r + 1] = make_uint4(0u, 0u, 0u, i);
    }
}
// (8) device-scope atomic adds at random words (no return value): the delta replicas
__global__ void cal_atomic_add4(uint32_t *__restrict__ p, uint32_t n, uint32_t word_mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&p[mix(i) & word_mask], 1u);
}

int main() {
    CHK(hipSetDevice(0));
    void *buf = nullptr, *hdr = nullptr, *mask = nullptr;
    uint32_t *sink = nullptr;
    CHK(hipMalloc(&buf, BUF));
    CHK(hipMemset(buf, 1, BUF));
    const uint32_t nslots_total = (uint32_t)(BUF / 4096);  // 524,288 slots
    CHK(hipMalloc(&hdr, (size_t)nslots_total * 32 + 128));
    CHK(hipMemset(hdr, 2, (size_t)nslots_total * 32 + 128));
    CHK(hipMalloc(&mask, (size_t)nslots_total * 4));
    CHK(hipMemset(mask, 3, (size_t)nslots_total * 4));
    CHK(hipMalloc((void **)&sink, 4));
    CHK(hipDeviceSynchronize());
    const dim3 g(256 * 8), b(256);
    const uint32_t nslots = nslots_total / 4;           // a quarter as many as there are slots, picked pseudo-randomly
    const uint32_t nrec = (uint32_t)(BUF / 32) / 8;     // 8 Mi records of 64 Mi
    const uint32_t rows = 32768, stride = 16384, ncols = 64;  // 2 GiB table of 16 Ki-word rows
    const uint32_t natom = 1u << 24;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(cal_read16_stream, g, b, 0, 0, (const uint4 *)buf, BUF / 16, sink);
        hipLaunchKernelGGL(cal_read4_stream, g, b, 0, 0, (const uint32_t *)buf, BUF / 4, sink);
        hipLaunchKernelGGL(cal_read_slots, g, b, 0, 0, (const uint32_t *)buf, (const uint4 *)hdr, (const uint32_t *)mask, nslots,
                           nslots_total - 1, sink);
        hipLaunchKernelGGL(cal_read_sector32, g, b, 0, 0, (const uint4 *)buf, nrec, (uint32_t)(BUF / 32) - 1, sink);
        hipLaunchKernelGGL(cal_read_column4, dim3(rows / 256), b, 0, 0, (const uint32_t *)buf, rows, stride, ncols, sink);
        hipLaunchKernelGGL(cal_write16_stream, g, b, 0, 0, (uint4 *)buf, BUF / 16);
        hipLaunchKernelGGL(cal_write_sector32, g, b, 0, 0, (uint4 *)buf, nrec, (uint32_t)(BUF / 32) - 1);
        hipLaunchKernelGGL(cal_atomic_add4, g, b, 0, 0, (uint32_t *)buf, natom, (uint32_t)(BUF / 4) - 1);
        CHK(hipDeviceSynchronize());
    }
    printf("{\"launches_each\": 3, \"requested_bytes_per_launch\": {"
           "\"cal_read16_stream\": {\"read\": %llu, \"write\": 0, \"pattern\": \"16 B per lane, coalesced, streaming\"}, "
           "\"cal_read4_stream\": {\"read\": %llu, \"write\": 0, \"pattern\": \"4 B per lane, coalesced, streaming\"}, "
           "\"cal_read_slots\": {\"read\": %llu, \"write\": 0, \"pattern\": \"per wave: a random 4 KiB slot + 96 B of headers + a 4 B mask word\"}, "
           "\"cal_read_sector32\": {\"read\": %llu, \"write\": 0, \"pattern\": \"one random aligned 32 B record per lane\"}, "
           "\"cal_read_column4\": {\"read\": %llu, \"write\": 0, \"pattern\": \"4 B per lane, 64 KiB apart (a table column); a 32 B sector per word = %llu\"}, "
           "\"cal_write16_stream\": {\"read\": 0, \"write\": %llu, \"pattern\": \"16 B per lane, coalesced, streaming\"}, "
           "\"cal_write_sector32\": {\"read\": 0, \"write\": %llu, \"pattern\": \"one random aligned 32 B record per lane (records may repeat)\"}, "
           "\"cal_atomic_add4\": {\"read\": 0, \"write\": %llu, \"pattern\": \"one 4 B atomic add per lane at a random word; a 32 B sector each = %llu\"}}}\n",
           (unsigned long long)BUF, (unsigned long long)BUF, (unsigned long long)nslots * (4096 + 96 + 4),
           (unsigned long long)nrec * 32, (unsigned long long)rows * ncols * 4, (unsigned long long)rows * ncols * 32,
           (unsigned long long)BUF, (unsigned long long)nrec * 32, (unsigned long long)natom * 4, (unsigned long long)natom * 32);
    return 0;
}
