// Fused ODEFunc right-hand side for H = 256:   Y = relu((A X) W^T + b)     (neural_dynamics.py:27-36)
//
// One persistent 512-thread workgroup per CU walks 64-row tiles.  Waves 4..7 ("producers") gather the
// NEXT tile's S = A X rows into one half of a double-buffered LDS tile while waves 0..3 ("consumers", one per
// SIMD) run the fp32 MFMA GEMM S W^T on the other half and write relu(. + b) - so the HBM-bound gather and
// the MFMA-bound GEMM overlap inside the CU, and S never travels to HBM (saves one write + one read of the
// N x H panel and a kernel boundary against the spmm -> linear pair).
//
//   LDS      : 2 x [64 rows][256 + 4 pad] fp32 = 133 120 B (of 160 KiB); pad 4 -> row m starts on 16-byte slot
//              (4 m) mod 64, so the consumers' ds_read_b128 of 16 distinct rows is conflict-free.
//   gather   : one row per producer wave at a time; the row's (col, val) pairs arrive by one coalesced load and
//              are broadcast with v_readlane; a neighbour row of X is one 1 KiB coalesced fetch; 8 in flight.
//   GEMM     : v_mfma_f32_32x32x2_f32 (exact fp32).  K is split in two halves: the instruction's k = 0 lanes
//              (0..31) walk k = 0..127 while its k = 1 lanes (32..63) walk k = 128..255, so every lane reads
//              4 consecutive k per ds_read_b128 / per 16-byte weight load.  Consumer wave w owns output
//              columns [64 w, 64 w + 64) (2 n-tiles) for both 32-row m-tiles: 4 accumulators, every weight
//              element is fetched once per tile (from L2: W is 256 KiB, shared by all CUs).
//   weights  : pre-packed once per call into MFMA operand order Wp[n-tile j][k-quad q][lane][4]
//              = W[32 j + (lane & 31)][128 (lane >> 5) + 4 q + 0..3], so a wave's B load is one 1 KiB access.
//   tiles    : each XCD owns a contiguous chunk of tiles and its 32 workgroups take them round-robin, so the
//              rows in flight on an XCD are ~2048 consecutive rows and their neighbours stay in that XCD's L2.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "kernels.h"
#include "split16.h"

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH = 256;
constexpr int kTileRows = 64;
constexpr int kLdS = kH + 4;                       // floats per LDS tile row
constexpr int kTileFloats = kTileRows * kLdS;
constexpr int kConsumerWaves = 4;

__global__ __launch_bounds__(256) void pack_weight_256_kernel(const float *__restrict__ W, float *__restrict__ Wp) {
    // Wp[((j * 32 + q) * 64 + lane) * 4 + e] = W[(32 j + (lane & 31)) * 256 + 128 (lane >> 5) + 4 q + e]
    const int idx = blockIdx.x * 256 + threadIdx.x;            // one float4 each: 8 * 32 * 64 = 16384
    if (idx >= 8 * 32 * 64) return;
    const int lane = idx & 63, q = (idx >> 6) & 31, j = idx >> 11;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(W + (32 * j + (lane & 31)) * kH + 128 * (lane >> 5) + 4 * q);
    reinterpret_cast<f32x4 *>(Wp)[idx] = v;
}

template <int U, bool HALO>
__device__ __forceinline__ void gather_batch(int c, float v, int i, const f32x4 *__restrict__ X,
                                             const f32x4 *__restrict__ Xh, int n_own, int lane, f32x4 &acc) {
    int cc[U];
    float vv[U];
    const f32x4 *pp[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        cc[q] = __builtin_amdgcn_readlane(c, i + q);
        vv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i + q));
        pp[q] = X;
        if (HALO && cc[q] >= n_own) { pp[q] = Xh; cc[q] -= n_own; }
    }
    f32x4 x[U];
#pragma unroll
    for (int q = 0; q < U; ++q) x[q] = pp[q][(size_t)cc[q] * 64 + lane];
#pragma unroll
    for (int q = 0; q < U; ++q) acc = vv[q] * x[q] + acc;
}

template <bool HALO>
__device__ __forceinline__ void gather_chunk(int c, float v, int cnt, const f32x4 *__restrict__ X,
                                             const f32x4 *__restrict__ Xh, int n_own, int lane, f32x4 &acc) {
    int i = 0;
    for (; i + 8 <= cnt; i += 8) gather_batch<8, HALO>(c, v, i, X, Xh, n_own, lane, acc);
    if (i + 4 <= cnt) { gather_batch<4, HALO>(c, v, i, X, Xh, n_own, lane, acc); i += 4; }
    if (i + 2 <= cnt) { gather_batch<2, HALO>(c, v, i, X, Xh, n_own, lane, acc); i += 2; }
    if (i < cnt) gather_batch<1, HALO>(c, v, i, X, Xh, n_own, lane, acc);
}

// Producer: rows [row0 + first, row0 + first + count) of S = A X into LDS tile `dst`.
template <bool HALO>
__device__ __forceinline__ void produce_rows(const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                             const float *__restrict__ val, const f32x4 *__restrict__ X,
                                             const f32x4 *__restrict__ Xh, int n_own, int n_rows, int row0, int first,
                                             int count, int lane, float *dst) {
    for (int rr = first; rr < first + count && rr < kTileRows; ++rr) {
        const int r = row0 + rr;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < n_rows) {
            const int j0 = rowptr[r], j1 = rowptr[r + 1];
            for (int jb = j0; jb < j1; jb += 64) {
                const int cnt = min(64, j1 - jb);
                int c = 0;
                float v = 0.f;
                if (lane < cnt) { c = colidx[jb + lane]; v = val[jb + lane]; }
                gather_chunk<HALO>(c, v, cnt, X, Xh, n_own, lane, acc);
            }
        }
        *reinterpret_cast<f32x4 *>(dst + rr * kLdS + 4 * lane) = acc;
    }
}

// Consumer wave `cw` (0..3): Y[row0 .. row0+64, 64 cw .. 64 cw + 64) = relu(S W^T + b) from LDS tile `src`.
__device__ __forceinline__ void consume_tile(const float *src, const f32x4 *__restrict__ Wp, const float *__restrict__ bias,
                                             float *__restrict__ Y, int n_rows, int row0, int cw, int lane, int relu) {
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const float *a0p = src + (lane & 31) * kLdS + 128 * (lane >> 5);
    const float *a1p = a0p + 32 * kLdS;
    const f32x4 *b0p = Wp + (size_t)(2 * cw) * 32 * 64 + lane;
    const f32x4 *b1p = b0p + 32 * 64;

    f32x4 b0 = b0p[0], b1 = b1p[0];
#pragma unroll 2
    for (int q = 0; q < 32; ++q) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p + 4 * q);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p + 4 * q);
        const f32x4 c0 = b0, c1 = b1;
        if (q + 1 < 32) { b0 = b0p[(q + 1) * 64]; b1 = b1p[(q + 1) * 64]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], c0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], c1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], c0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], c1[e], acc[1][1], 0, 0, 0);
        }
    }
    // D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][n = lane & 31]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = 64 * cw + 32 * n + (lane & 31);
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < n_rows) {
                    float o = acc[m][n][r] + bv;
                    if (relu) o = relu_nan(o);
                    __builtin_nontemporal_store(o, &Y[(size_t)row * kH + col]);
                }
            }
    }
}

template <int NPROD, bool HALO>
__global__ __launch_bounds__(256 + 64 * NPROD) void rhs_fused_256_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colidx, const float *__restrict__ val,
    const float *__restrict__ Xf, const float *__restrict__ Xhf, int n_own, const float *__restrict__ Wpf,
    const float *__restrict__ bias, float *__restrict__ Y, int n_rows, int n_tiles, int relu) {
    __shared__ __attribute__((aligned(16))) float s_tile[2 * kTileFloats];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool producer = wave >= kConsumerWaves;
    const f32x4 *X = reinterpret_cast<const f32x4 *>(Xf);
    const f32x4 *Xh = reinterpret_cast<const f32x4 *>(Xhf);
    const f32x4 *Wp = reinterpret_cast<const f32x4 *>(Wpf);

    // tiles of this workgroup: XCD x owns tiles [x * chunk, (x+1) * chunk); its workgroups take them round-robin
    const int xcd = blockIdx.x % kXcds;
    const int wg = blockIdx.x / kXcds;
    const int wgs_per_xcd = gridDim.x / kXcds;
    const int chunk = (n_tiles + kXcds - 1) / kXcds;
    const int t_lo = xcd * chunk, t_hi = min(n_tiles, t_lo + chunk);
    constexpr int kRowsPerProducer = (kTileRows + NPROD - 1) / NPROD;      // 12 producers: 6 rows, the last 2 idle

    int t = t_lo + wg;
    if (t >= t_hi) return;                                     // whole workgroup: uniform
    if (producer)
        produce_rows<HALO>(rowptr, colidx, val, X, Xh, n_own, n_rows, t * kTileRows,
                           (wave - kConsumerWaves) * kRowsPerProducer, kRowsPerProducer, lane, s_tile);
    __syncthreads();
    int buf = 0;
    for (; t < t_hi; t += wgs_per_xcd) {
        const int tn = t + wgs_per_xcd;
        if (producer) {
            if (tn < t_hi)
                produce_rows<HALO>(rowptr, colidx, val, X, Xh, n_own, n_rows, tn * kTileRows,
                                   (wave - kConsumerWaves) * kRowsPerProducer, kRowsPerProducer, lane,
                                   s_tile + (buf ^ 1) * kTileFloats);
        } else {
            consume_tile(s_tile + buf * kTileFloats, Wp, bias, Y, n_rows, t * kTileRows, wave, lane, relu);
        }
        __syncthreads();
        buf ^= 1;
    }
}

static int env_int2(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

int rhs_fused_supported(int H, uint32_t flags) {
    static const int enabled = env_int2("NDCN_RHS_FUSED", 1);
    if (!enabled) return 0;
    if (flags & (NDCN_F_NO_GRAPH | NDCN_F_NO_CONTROL)) return 0;
    return H == kH ? 1 : 0;
}

// packed fp32 weights (256 KiB) followed by the split fp16 weights (two planes, 256 KiB, + 256 unscale factors; sized for three planes)
int64_t rhs_fused_work_bytes(int H) { return (int64_t)H * H * sizeof(float) + (int64_t)H * H * 3 * 2; }

// Split weights for the fp16 consumers (split16.h): every row n of the B operand (W[n][:]; TRANSPOSED: W[:][n]) gets a power-of-two
// scale of its own that brings its largest magnitude into [2^14, 2^15), then every scaled weight becomes two fp16 pieces (round to
// nearest, then the exact remainder rounded to nearest), MFMA 32x32x16 B-operand order:
//   Wh[(((j * 16 + s) * 2 + p) * 64 + lane) * 8 + e] = piece p of B[32 j + (lane & 31)][16 s + 8 (lane >> 5) + e];
// behind the planes: float unscale[256] = 1 / scale of row n (the consumers multiply output column n back by it).
// A block packs four k-steps of ONE n-tile j: it needs the maxima of rows 32 j .. 32 j + 31 only (8 threads per row, 32 values each).
// Range guard (round 6): the two-piece split keeps min(21, 38 - e) bits of an element 2^-e below its row's largest magnitude
// (split16.h), i.e. elements more than 2^kS16GuardBits below it can - when the rest of the row meets zeros of S - put the output
// outside 2e-6 sum |s w|.  Behind the scan that forms the row maximum a second one counts the NON-ZERO elements below that mark; a row
// with kS16GuardCount or more of them (a row that spans decades - not the one or two stragglers every random matrix has: 5 % of the
// nn.Linear default initialisations and 15 % of Gaussian matrices hold an element 2^-19 below its row's maximum, none holds four in
// one row) is reported per n-tile in 8 words behind the 256 unscale factors, and pack_weight_256 hands the verdict to the host, which
// routes such weights to the fp32 matrix-core kernels (rhs.hip: NDCN_PATH_EXACT32; nn.Linear in fp32, neural_dynamics.py:33).
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void pack_weight_256_f16_kernel(const float *__restrict__ W, _Float16 *__restrict__ Wh,
                                                                  float *__restrict__ tail) {
    __shared__ float s_scale[32];
    __shared__ int s_wide;
    const int idx = blockIdx.x * 256 + threadIdx.x;            // one (j, s, lane): 8 * 16 * 64 = 8192
    const int lane = idx & 63, s = (idx >> 6) & 15, j = idx >> 10;
    if (threadIdx.x == 0) s_wide = 0;
    __syncthreads();
    {
        const int r = threadIdx.x >> 3, part = threadIdx.x & 7, n = 32 * j + r;
        unsigned m = 0;
        for (int k = 32 * part; k < 32 * part + 32; ++k) {
            const unsigned b = __builtin_bit_cast(unsigned, TRANSPOSED ? W[k * kH + n] : W[n * kH + k]) & 0x7fffffffu;
            m = b > m ? b : m;
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, off, 64);
            m = o > m ? o : m;
        }
        // (all eight threads of the row hold its maximum now) non-zero elements below the guarantee, exponents compared
        int below = 0;
        if ((blockIdx.x & 3) == 0) {
            for (int k = 32 * part; k < 32 * part + 32; ++k) {
                const unsigned b = __builtin_bit_cast(unsigned, TRANSPOSED ? W[k * kH + n] : W[n * kH + k]) & 0x7fffffffu;
                below += (b != 0u && (int)(m >> 23) - (int)(b >> 23) > kS16GuardBits) ? 1 : 0;
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) below += __shfl_xor(below, off, 64);
        }
        if (part == 0) {
            unsigned sb, ub;
            s16_scale_bits(m, sb, ub);
            s_scale[r] = __builtin_bit_cast(float, sb);
            if ((blockIdx.x & 3) == 0) tail[n] = __builtin_bit_cast(float, ub);
            // finite rows only (a NaN / Inf row is non-finite on every route)
            if ((m >> 23) != 255u && below >= kS16GuardCount) atomicOr(&s_wide, 1);
        }
    }
    __syncthreads();
    if ((blockIdx.x & 3) == 0 && threadIdx.x == 0) reinterpret_cast<int *>(tail)[256 + j] = s_wide;
    const int row = 32 * j + (lane & 31), col = 16 * s + 8 * (lane >> 5);
    const float sc = s_scale[lane & 31];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float w = (TRANSPOSED ? W[(col + e) * kH + row] : W[row * kH + col + e]) * sc;
        const _Float16 g0 = (_Float16)w;
        const _Float16 g1 = (_Float16)(w - (float)g0);
        Wh[((((size_t)j * 16 + s) * 2 + 0) * 64 + lane) * 8 + e] = g0;
        Wh[((((size_t)j * 16 + s) * 2 + 1) * 64 + lane) * 8 + e] = g1;
    }
}

// ---- which packed images hold wide-range weights: decided when the image is packed, asked at every launch that reads it -----------
namespace {
std::mutex g_wide_mu;
std::unordered_map<const void *, bool> g_wide;          // key: the packed image (a caller's scratch: few, long-lived)
std::atomic<int> g_guard_override{-1};                  // ndcn_set_range_guard: -1 = the environment's default
bool range_guard_on() {
    static const bool dflt = [] { const char *e = getenv("NDCN_RANGE_GUARD"); return !(e && e[0] == '0'); }();
    const int o = g_guard_override.load(std::memory_order_relaxed);
    return o < 0 ? dflt : o != 0;
}
}  // namespace

int set_range_guard(int on) {
    const int prev = range_guard_on() ? 1 : 0;
    g_guard_override.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed);
    if (!range_guard_on()) {                                 // verdicts of the guarded time do not outlive it
        std::lock_guard<std::mutex> lk(g_wide_mu);
        g_wide.clear();
    }
    return prev;
}

bool weights_wide_range(const void *Wp) {
    if (!range_guard_on()) return false;
    std::lock_guard<std::mutex> lk(g_wide_mu);
    auto it = g_wide.find(Wp);
    return it != g_wide.end() && it->second;
}

// 32 bytes back to the host, once per packed image (a solve packs once; callers that keep the image pass NDCN_F_PACKED).  Not while the
// stream is being captured: a captured pack keeps the verdict of the image's last eager pack.
static int learn_verdict(const void *image, const float *tail, hipStream_t st) {
    if (!range_guard_on()) return NDCN_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return NDCN_OK;
    int h[8];
    NDCN_HIP(hipMemcpyAsync(h, reinterpret_cast<const int *>(tail) + 256, sizeof(h), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    bool wide = false;
    for (int q = 0; q < 8; ++q) wide = wide || h[q] != 0;
    std::lock_guard<std::mutex> lk(g_wide_mu);
    if (g_wide.size() > 4096) g_wide.clear();                    // (images come and go with their solvers: bounded, re-learnt at the next pack)
    g_wide[image] = wide;
    return NDCN_OK;
}

int pack_weight_256(const float *W, float *Wp, hipStream_t st) {
    hipLaunchKernelGGL(pack_weight_256_kernel, dim3(64), dim3(256), 0, st, W, Wp);
    float *tail = reinterpret_cast<float *>(reinterpret_cast<char *>(Wp + kH * kH) + kS16Bytes);
    hipLaunchKernelGGL(pack_weight_256_f16_kernel<false>, dim3(32), dim3(256), 0, st, W, reinterpret_cast<_Float16 *>(Wp + kH * kH), tail);
    NDCN_LAUNCH_CHECK();
    return learn_verdict(Wp, tail, st);
}

// Wq <- the fp16 planes of W^T + the 256 per-row unscale factors + the range guard's 8 words (kS16Bytes + kS16TailBytes + kS16GuardBytes)
int pack_weight_256_t16(const float *W, void *Wq, hipStream_t st) {
    float *tail = reinterpret_cast<float *>(static_cast<char *>(Wq) + kS16Bytes);
    hipLaunchKernelGGL(pack_weight_256_f16_kernel<true>, dim3(32), dim3(256), 0, st, W, static_cast<_Float16 *>(Wq), tail);
    NDCN_LAUNCH_CHECK();
    return learn_verdict(Wq, tail, st);                          // (rows of W^T = columns of W: the backward's gS = gZ W, linear_bwd.hip)
}

// Wp: packed weights (pack_weight_256).  X, Y, W 16-byte aligned.
int rhs_fused_packed_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *Wp,
                         const float *b, float *Y, uint32_t flags, hipStream_t st) {
    const int n_rows = (int)A->n_rows;
    if (n_rows == 0) return NDCN_OK;
    const int n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    static const int nprod = env_int2("NDCN_RHS_PRODUCERS", 8);   // measured: 4 -> 2.3 ms, 8 -> 1.84 ms, 12 -> 1.88 ms (1M grid)
    int per_xcd = kCus / kXcds;                                  // one workgroup per CU
    const int need = (n_tiles + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds);
    const int relu = (flags & NDCN_F_RELU) ? 1 : 0;
    ProfScope prof(PROF_RHS_FUSED, st,
                   8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * kH * (double)(A->n_rows + A->n_cols) + 4.0 * kH * kH,
                   2.0 * A->nnz * kH + 2.0 * (double)A->n_rows * kH * kH);
#define NDCN_FUSED(NP)                                                                                              \
    do {                                                                                                            \
        if (Xh)                                                                                                     \
            hipLaunchKernelGGL((rhs_fused_256_kernel<NP, true>), grid, dim3(256 + 64 * NP), 0, st, A->rowptr,       \
                               A->colidx, A->val, X, Xh, (int)n_own, Wp, b, Y, n_rows, n_tiles, relu);              \
        else                                                                                                        \
            hipLaunchKernelGGL((rhs_fused_256_kernel<NP, false>), grid, dim3(256 + 64 * NP), 0, st, A->rowptr,      \
                               A->colidx, A->val, X, Xh, (int)n_own, Wp, b, Y, n_rows, n_tiles, relu);              \
    } while (0)
    if (nprod >= 12) NDCN_FUSED(12);
    else if (nprod >= 8) NDCN_FUSED(8);
    else NDCN_FUSED(4);
#undef NDCN_FUSED
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

int rhs_fused_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, const float *W, const float *b,
                  float *Y, float *work, int H, uint32_t flags, hipStream_t st) {
    if (H != kH || !work) { set_error("rhs_fused: needs H == 256 and a %d-byte scratch", kH * kH * 4); return NDCN_EINVAL; }
    if (!(aligned16(X) && aligned16(Y) && aligned16(W) && aligned16(work) && (!Xh || aligned16(Xh)))) {
        set_error("rhs_fused: panels must be 16-byte aligned");
        return NDCN_EINVAL;
    }
    int rc = pack_weight_256(W, work, st);
    if (rc) return rc;
    return rhs_fused_packed_f32(A, X, Xh, n_own, work, b, Y, flags, st);
}

}  // namespace ndcn
