// Internal helpers shared by the kernels of libndcn_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ndcn_hip.h"
#include "prof.h"

namespace ndcn {

void set_error(const char *fmt, ...);

#define NDCN_CHECK_ARG(cond, msg)                                         \
    do {                                                                  \
        if (!(cond)) {                                                    \
            ::ndcn::set_error("%s: %s", __func__, msg);                   \
            return NDCN_EINVAL;                                           \
        }                                                                 \
    } while (0)

#define NDCN_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::ndcn::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
            return NDCN_EHIP;                                                              \
        }                                                                                  \
    } while (0)

#define NDCN_LAUNCH_CHECK()                                                                   \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            ::ndcn::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e_)); \
            return NDCN_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kXcds = 8;        // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)
constexpr int kCus = 256;
// The persistent grids (one workgroup per CU, XCD = blockIdx % 8) are LAID OUT for the whole chip; on a partitioned or smaller device
// (CPX / DPX modes report 32 / 128 compute units) they stay correct - every workgroup strides over the work - and merely
// oversubscribe.  Kernels that NEED all their workgroups co-resident (the column sweep's progress words, spmm_sweep.hip) ask this
// first and refuse otherwise (their callers fall back to the row kernels).  Cached per device.
inline bool device_is_whole_chip() {
    static std::atomic<int> known[64];                     // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    int k = known[dev].load(std::memory_order_relaxed);
    if (k == 0) {
        int cus = 0;
        k = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == kCus) ? 1 : 2;
        known[dev].store(k, std::memory_order_relaxed);
    }
    return k == 1;
}

// torch's relu / max propagate NaN (F.relu, neural_dynamics.py:36; torch.max, misc.py:149): v_max_f32 (fmaxf) does not - it
// returns the other operand, which would turn a NaN born inside the right-hand side into K = 0 and hide it from the
// solver's finiteness assertion (dopri5.py:101-102).
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Contiguous-chunk-per-XCD remap of a 1-D grid (bijective for any grid size): blocks that the
// dispatcher places on one XCD (b % 8 equal) get CONSECUTIVE logical ids, so neighbouring row blocks
// share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk / kXcds, r = nblk % kXcds;
    const int xcd = b % kXcds, k = b / kXcds;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// grid size for streaming elementwise kernels: enough blocks to fill the chip, grid-stride the rest
inline int stream_grid(int64_t n_items, int block) {
    int64_t g = (n_items + block - 1) / block;
    const int64_t cap = (int64_t)kCus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// Streaming elementwise kernels WITHOUT a reduction: one item per thread, as many workgroups as that takes.  Measured
// (tools/micro/copy_lab.hip, 1 GiB float4 copy): one float4 per thread with non-temporal accesses 6.49 TB/s; the same
// loop body on a persistent grid of 2048-16384 workgroups 4.1-5.7 TB/s - a grid-stride loop makes every wave of the chip
// alternate between a burst of loads and a burst of stores in step, while the dispatcher's stream of short-lived
// workgroups keeps reads and writes mixed.  NDCN_STREAM_FULL=0 restores the capped grid (A/B).
int stream_grid_full(int64_t n_items, int block);

// A per-kernel one-off (hipFuncSetAttribute is per DEVICE): true exactly once per device of this process for the flag word it
// is handed - a model on cuda:1, a thread per rank (round-4 advisor: a process-wide `static bool` skipped the second device).
inline bool once_per_device(std::atomic<unsigned long long> &seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;       // unknown: set the attribute again (cheap, idempotent)
    const unsigned long long bit = 1ull << dev;
    return (seen.fetch_or(bit) & bit) == 0;
}

}  // namespace ndcn
