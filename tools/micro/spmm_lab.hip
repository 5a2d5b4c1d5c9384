// SpMM laboratory (tuning aid, not product code): standalone variants of the H = 256 CSR SpMM on the metric's case
// (1000 x 1000 8-neighbour lattice, 9 entries per row) timed with HIP events and checked bit for bit against a
// sequential-fma CPU sum.  Baselines come from the product library through its C ABI.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/spmm_lab tools/micro/spmm_lab.hip \
//         -Indcn_amd/csrc -Lndcn_amd -l:libndcn_hip.so -Wl,-rpath,'$ORIGIN/../../ndcn_amd'
//   tools/micro/spmm_lab [side] [reps]
//
// Variant family "pipe": ONE persistent workgroup per CU walks groups of R rows.  For each group the DISTINCT
// neighbour rows (the union, padded to CAP = CAPW * W slots) are brought into an LDS ring of NBUF buffers by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip), NBUF - 1 groups ahead of the group being summed; the sums read
// LDS (ds_read_b128) only.  One workgroup barrier per group; the DMA completion is awaited with a counted vmcnt.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ndcn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kXcds = 8;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 1 KiB row, global -> LDS, no VGPR data path: LDS address = M0 (wave-uniform) + lane * 16
__device__ __forceinline__ void dma_row(const float *row_base /*uniform*/, unsigned lds_byte /*uniform*/, int lane_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_byte), "s"(row_base) : "memory");
}

__device__ __forceinline__ f32x4 fma4(float s, f32x4 x, f32x4 a) {
    return (f32x4){fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w)};
}

// U entries j .. j+U-1 of one row out of the LDS stage
template <int U>
__device__ __forceinline__ void lds_chunk(const int *__restrict__ lidx, const float *__restrict__ val, int j, const f32x4 *buf,
                                          int lane, f32x4 &acc) {
    int li[U];
    float vv[U];
#pragma unroll
    for (int q = 0; q < U; ++q) { li[q] = lidx[j + q]; vv[q] = val[j + q]; }
    f32x4 x[U];
#pragma unroll
    for (int q = 0; q < U; ++q) x[q] = buf[li[q] * 64 + lane];
#pragma unroll
    for (int q = 0; q < U; ++q) acc = fma4(vv[q], x[q], acc);
}

// DBG bit 0: no stores; bit 1: no DMA (LDS garbage); bit 2: no LDS reads / sums
template <int R, int W, int CAPW, int NBUF, int DBG>
__global__ __launch_bounds__(64 * W) void spmm_pipe(const int *__restrict__ rowptr, const int *__restrict__ lidx,
                                                     const float *__restrict__ val, const int *__restrict__ grp_rows,
                                                     const int *__restrict__ ucols, const float *__restrict__ Xf,
                                                     float *__restrict__ Yf, int n_groups) {
    constexpr int CAP = CAPW * W;
    constexpr int RPW = R / W;
    constexpr int D = NBUF - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_off = lane * 16;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    f32x4 *Y = reinterpret_cast<f32x4 *>(Yf);

    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wpx = gridDim.x / kXcds;
    const int chunk = (n_groups + kXcds - 1) / kXcds;
    const int g_lo = xcd * chunk, g_hi = min(n_groups, g_lo + chunk);
    const int g0 = g_lo + wg;
    if (g0 >= g_hi) return;
    const int my = (g_hi - g0 + wpx - 1) / wpx;

    auto stage = [&](int g, int b) {
        const int *uc = ucols + (size_t)g * CAP + wave * CAPW;
        int cc[CAPW];
#pragma unroll
        for (int k = 0; k < CAPW; ++k) cc[k] = uc[k];
#pragma unroll
        for (int k = 0; k < CAPW; ++k)
            if (!(DBG & 2)) dma_row(Xf + (size_t)cc[k] * 256, lds_base + (unsigned)((b * CAP + wave * CAPW + k) * 1024), lane_off);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < my) stage(g0 + d * wpx, d);

    int b = 0;
    for (int it = 0; it < my; ++it) {
        const int g = g0 + it * wpx;
        // DMA(it) must have landed.  Younger vector-memory ops of this wave, in steady state: the DMA of the next
        // D - 1 groups and the stores of the last D groups (vector memory completes in order).
        if (it >= D && it + D <= my) wait_vmcnt<(DBG & 2 ? 0 : (D - 1) * CAPW) + (DBG & 1 ? 0 : D * RPW)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + D < my) { int bn = b + D; if (bn >= NBUF) bn -= NBUF; stage(g + D * wpx, bn); }
        const f32x4 *buf = reinterpret_cast<const f32x4 *>(lds) + b * CAP * 64;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = grp_rows[(size_t)g * R + wave + W * q];
            int j = rowptr[r];
            const int j1 = rowptr[r + 1];
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!(DBG & 4)) {
                for (; j1 - j >= 8; j += 8) lds_chunk<8>(lidx, val, j, buf, lane, acc);
                const int m = j1 - j;
                if (m & 4) { lds_chunk<4>(lidx, val, j, buf, lane, acc); j += 4; }
                if (m & 2) { lds_chunk<2>(lidx, val, j, buf, lane, acc); j += 2; }
                if (m & 1) lds_chunk<1>(lidx, val, j, buf, lane, acc);
            } else {
                acc.x = (float)j1;
            }
            if (!(DBG & 1)) __builtin_nontemporal_store(acc, &Y[(size_t)r * 64 + lane]);
            else if (acc.x == 1.2345e-30f) Y[(size_t)r * 64 + lane] = acc;
        }
        if (++b == NBUF) b = 0;
    }
}


// ------------------------------------------------------------------------------------------------
// Variant family "rec": every byte a group needs arrives by LDS-DMA at arithmetic addresses - no load of streamed
// index data stands between a wave and its work.  Per group the plan holds one fixed-size RECORD
//   words [0, CAP)            the union's column ids (padded by repetition)
//   words [CAP, CAP + 2R)     per row {row id, cnt | ofs << 16}   (cnt 0xffff: the group does not fit - direct gather)
//   words [CAP + 2R, ...)     the rows' entries as (slot, value) pairs, row after row
// Roles: WD DMA waves (record of group it+2D, then the union rows of group it+D whose column ids they read out of the
// record that landed earlier; their only waits are counted vmcnt on their own DMA) and WC compute waves (record +
// rows out of LDS, plain stores; the compiler's own wait insertion is exact for them).  One barrier per group.
template <int R, int WC, int WD, int CAP, int D, int RECW, int DBG>
__global__ __launch_bounds__(64 * (WC + WD)) void spmm_rec(const int *__restrict__ rec, const float *__restrict__ Xf,
                                                           float *__restrict__ Yf, int n_groups) {
    constexpr int NBUF = D + 1, NREC = 2 * D + 1, CAPD = CAP / WD, RPW = R / WC, E0 = CAP + 2 * R;
    static_assert(CAP % WD == 0 && R % WC == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_off = lane * 16;
    const unsigned lds_x = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    const unsigned lds_r = lds_x + NBUF * CAP * 1024;
    const int *rbuf = reinterpret_cast<const int *>(lds) + NBUF * CAP * 256;

    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wpx = gridDim.x / kXcds;
    const int chunk = (n_groups + kXcds - 1) / kXcds;
    const int g_lo = xcd * chunk, g_hi = min(n_groups, g_lo + chunk);
    const int g0 = g_lo + wg;
    if (g0 >= g_hi) return;
    const int my = (g_hi - g0 + wpx - 1) / wpx;

    if (wave < WD) {
        const int d = wave;
        auto dma_rec = [&](int it) {
            if (d < RECW && it < my)
                dma_row(reinterpret_cast<const float *>(rec + ((size_t)(g0 + it * wpx) * RECW + d) * 256),
                        lds_r + (unsigned)(((it % NREC) * RECW + d) * 1024), lane_off);
        };
        auto dma_x = [&](int it) {
            const int *r = rbuf + (it % NREC) * RECW * 256 + d * CAPD;
            int cc[CAPD];
#pragma unroll
            for (int k = 0; k < CAPD; ++k) cc[k] = __builtin_amdgcn_readfirstlane(r[k]);
#pragma unroll
            for (int k = 0; k < CAPD; ++k)
                dma_row(Xf + (size_t)cc[k] * 256, lds_x + (unsigned)(((it % NBUF) * CAP + d * CAPD + k) * 1024), lane_off);
        };
        for (int it = 0; it < 2 * D; ++it) dma_rec(it);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < D; ++it)
            if (it < my) dma_x(it);
        for (int it = 0; it < my; ++it) {
            // the union rows of group `it` (and the record issued just ahead of them) must have landed; younger ops of
            // this wave in steady state: D - 1 iterations of DMA
            if (it >= D && it + 2 * D <= my) {
                if (d < RECW) wait_vmcnt<(D - 1) * (CAPD + 1)>();
                else wait_vmcnt<(D - 1) * CAPD>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            dma_rec(it + 2 * D);
            if (it + D < my) dma_x(it + D);
        }
        wait_vmcnt<0>();
    } else {
        const int c = wave - WD;
        f32x4 *Y = reinterpret_cast<f32x4 *>(Yf);
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < my; ++it) {
            __builtin_amdgcn_s_barrier();
            const int *r = rbuf + (it % NREC) * RECW * 256;
            const f32x4 *xb = reinterpret_cast<const f32x4 *>(lds) + (it % NBUF) * CAP * 64;
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int i = c + WC * q;
                const int row = __builtin_amdgcn_readfirstlane(r[CAP + 2 * i]);
                const int meta = __builtin_amdgcn_readfirstlane(r[CAP + 2 * i + 1]);
                const int cnt = meta & 0xffff, ofs = meta >> 16;
                int es = 0;
                float ev = 0.f;
                if (lane < cnt) { es = r[E0 + 2 * (ofs + lane)]; ev = __builtin_bit_cast(float, r[E0 + 2 * (ofs + lane) + 1]); }
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                auto chunk_u = [&](auto U_, int base) {
                    constexpr int U = decltype(U_)::value;
                    int sl[U];
                    float vv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        sl[u] = __builtin_amdgcn_readlane(es, base + u);
                        vv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), base + u));
                    }
                    f32x4 x[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) x[u] = xb[sl[u] * 64 + lane];
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = fma4(vv[u], x[u], acc);
                };
                if (!(DBG & 4)) {
                    int j = 0;
                    for (; cnt - j >= 8; j += 8) chunk_u(std::integral_constant<int, 8>{}, j);
                    const int m = cnt - j;
                    if (m & 4) { chunk_u(std::integral_constant<int, 4>{}, j); j += 4; }
                    if (m & 2) { chunk_u(std::integral_constant<int, 2>{}, j); j += 2; }
                    if (m & 1) chunk_u(std::integral_constant<int, 1>{}, j);
                }
                if (!(DBG & 1)) __builtin_nontemporal_store(acc, &Y[(size_t)row * 64 + lane]);
                else if (acc.x == 1.2345e-30f) Y[(size_t)row * 64 + lane] = acc;
            }
        }
    }
}

// plain persistent row copy (one row per wave, 8 in flight): the ceiling of "read 1 KiB rows, write 1 KiB rows"
__global__ __launch_bounds__(256) void row_copy(const f32x4 *__restrict__ X, f32x4 *__restrict__ Y, int n_rows) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int r = w; r < n_rows; r += nw * 4) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) if (r + q * nw < n_rows) v[q] = __builtin_nontemporal_load(&X[(size_t)(r + q * nw) * 64 + lane]);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (r + q * nw < n_rows) __builtin_nontemporal_store(v[q], &Y[(size_t)(r + q * nw) * 64 + lane]);
    }
}

// ------------------------------------------------------------------------------------------------ host
struct Csr { int n; std::vector<int> rp, ci; std::vector<float> va; };

static Csr lattice(int S) {
    Csr A; A.n = S * S; A.rp.assign(A.n + 1, 0);
    std::vector<int> deg(A.n, 0);
    for (int x = 0; x < S; ++x) for (int y = 0; y < S; ++y) {
        int d = 0;
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) {
            if (!dx && !dy) continue;
            const int u = x + dx, v = y + dy;
            if (u >= 0 && u < S && v >= 0 && v < S) ++d;
        }
        deg[x * S + y] = d;
    }
    for (int x = 0; x < S; ++x) for (int y = 0; y < S; ++y) {
        const int i = x * S + y;
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) {
            const int u = x + dx, v = y + dy;
            if (u < 0 || u >= S || v < 0 || v >= S) continue;
            const int j = u * S + v;
            A.ci.push_back(j);
            A.va.push_back(i == j ? 1.f : -1.f / std::sqrt((float)deg[i] * (float)deg[j]));
        }
        A.rp[i + 1] = (int)A.ci.size();
    }
    return A;
}

static Csr identity(int n) {
    Csr A; A.n = n; A.rp.resize(n + 1); A.ci.resize(n); A.va.assign(n, 1.f);
    for (int i = 0; i <= n; ++i) A.rp[i] = i;
    for (int i = 0; i < n; ++i) A.ci[i] = i;
    return A;
}

struct PipePlan { int R, W, CAPW, n_groups; std::vector<int> grp_rows, ucols, lidx; double loads_per_row; };

// order: the walk order of the rows; consecutive R entries form a group
static bool build_plan(const Csr &A, const std::vector<int> &order, int R, int W, int CAPW, PipePlan &P) {
    const int CAP = CAPW * W;
    P.R = R; P.W = W; P.CAPW = CAPW; P.n_groups = (A.n + R - 1) / R;
    P.grp_rows.assign((size_t)P.n_groups * R, 0);
    P.ucols.assign((size_t)P.n_groups * CAP, 0);
    P.lidx.assign(A.ci.size() + 32, 0);
    std::vector<int> u;
    size_t total = 0;
    for (int g = 0; g < P.n_groups; ++g) {
        u.clear();
        for (int i = 0; i < R; ++i) {
            const int r = order[std::min(A.n - 1, g * R + i)];
            P.grp_rows[(size_t)g * R + i] = r;
            for (int j = A.rp[r]; j < A.rp[r + 1]; ++j) u.push_back(A.ci[j]);
        }
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if ((int)u.size() > CAP) { fprintf(stderr, "group %d: union %zu > cap %d\n", g, u.size(), CAP); return false; }
        total += u.size();
        for (int s = 0; s < CAP; ++s) P.ucols[(size_t)g * CAP + s] = u[std::min<size_t>(s, u.size() - 1)];
        for (int i = 0; i < R; ++i) {
            const int r = P.grp_rows[(size_t)g * R + i];
            for (int j = A.rp[r]; j < A.rp[r + 1]; ++j)
                P.lidx[j] = (int)(std::lower_bound(u.begin(), u.end(), A.ci[j]) - u.begin());
        }
    }
    P.loads_per_row = (double)total / A.n;
    return true;
}


struct RecPlan { int n_groups; std::vector<int> rec; double loads_per_row; int unfit; };
static bool build_rec(const Csr &A, const std::vector<int> &order, int R, int CAP, int RECW, RecPlan &P) {
    const int words = RECW * 256, E0 = CAP + 2 * R, ecap = (words - E0) / 2;
    P.n_groups = (A.n + R - 1) / R;
    P.rec.assign((size_t)P.n_groups * words, 0);
    P.unfit = 0;
    std::vector<int> u;
    size_t total = 0;
    for (int g = 0; g < P.n_groups; ++g) {
        int *rec = &P.rec[(size_t)g * words];
        u.clear();
        int rows[64];
        size_t ne = 0;
        for (int i = 0; i < R; ++i) {
            rows[i] = order[std::min(A.n - 1, g * R + i)];
            for (int j = A.rp[rows[i]]; j < A.rp[rows[i] + 1]; ++j) u.push_back(A.ci[j]);
            ne += A.rp[rows[i] + 1] - A.rp[rows[i]];
        }
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if ((int)u.size() > CAP || (int)ne > ecap) { fprintf(stderr, "group %d does not fit (union %zu, entries %zu)\n", g, u.size(), ne); ++P.unfit; return false; }
        total += u.size();
        for (int s = 0; s < CAP; ++s) rec[s] = u[std::min<size_t>(s, u.size() - 1)];
        int ofs = 0;
        for (int i = 0; i < R; ++i) {
            const int r = rows[i], cnt = A.rp[r + 1] - A.rp[r];
            rec[CAP + 2 * i] = r;
            rec[CAP + 2 * i + 1] = cnt | (ofs << 16);
            for (int j = A.rp[r]; j < A.rp[r + 1]; ++j) {
                rec[E0 + 2 * (ofs + j - A.rp[r])] = (int)(std::lower_bound(u.begin(), u.end(), A.ci[j]) - u.begin());
                memcpy(&rec[E0 + 2 * (ofs + j - A.rp[r]) + 1], &A.va[j], 4);
            }
            ofs += cnt;
        }
    }
    P.loads_per_row = (double)total / A.n;
    return true;
}

template <class T> static T *dev(const std::vector<T> &v, size_t pad = 0) {
    T *p; HIPCHECK(hipMalloc(&p, (v.size() + pad) * sizeof(T) + 64));
    HIPCHECK(hipMemset(p, 0, (v.size() + pad) * sizeof(T) + 64));
    HIPCHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

static float timeit(int reps, const std::function<void()> &fn) {
    for (int i = 0; i < 3; ++i) fn();
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t a, b; HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
    HIPCHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) fn();
    HIPCHECK(hipEventRecord(b, 0));
    HIPCHECK(hipEventSynchronize(b));
    float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b));
    HIPCHECK(hipGetLastError());
    return ms / reps;
}

static int check(const Csr &A, const std::vector<float> &X, const float *dY, const char *name) {
    std::vector<float> row(256);
    int bad = 0;
    const int step = std::max(1, A.n / 3001);
    for (int r = 0; r < A.n; r += step) {
        HIPCHECK(hipMemcpy(row.data(), dY + (size_t)r * 256, 1024, hipMemcpyDeviceToHost));
        for (int c = 0; c < 256; ++c) {
            float s = 0.f;
            for (int j = A.rp[r]; j < A.rp[r + 1]; ++j) s = fmaf(A.va[j], X[(size_t)A.ci[j] * 256 + c], s);
            if (memcmp(&s, &row[c], 4) != 0) { if (bad < 3) fprintf(stderr, "  %s: row %d col %d: %g vs %g\n", name, r, c, row[c], s); ++bad; }
        }
    }
    return bad;
}

int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 1000;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const Csr A = lattice(S);
    const int n = A.n;
    const size_t nnz = A.ci.size();
    const double alg = 8.0 * nnz + 4.0 * (n + 1) + 8.0 * 256 * (double)n;
    printf("lattice %d x %d: n %d nnz %zu  algorithmic bytes %.3f GB  (target 0.50 of 8 TB/s: %.3f ms)\n", S, S, n, nnz, alg / 1e9,
           alg / 4e12 * 1e3);
    std::vector<float> X((size_t)n * 256);
    unsigned s = 12345u;
    for (auto &x : X) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) / 16777216.f; }
    float *dX = dev(X), *dY;
    HIPCHECK(hipMalloc(&dY, (size_t)n * 1024));
    int *d_rp = dev(A.rp), *d_ci = dev(A.ci, 32);
    float *d_va = dev(A.va, 32);

    auto report = [&](const char *name, float ms, double bytes, int bad, const char *note) {
        printf("{\"variant\": \"%s\", \"ms\": %.4f, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f, \"mismatches\": %d, \"note\": \"%s\"}\n", name, ms,
               bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, bad, note);
        fflush(stdout);
    };

    // ---- ceilings
    {
        float ms = timeit(reps, [&] { hipLaunchKernelGGL(row_copy, dim3(256 * 8), dim3(256), 0, 0, (const f32x4 *)dX, (f32x4 *)dY, n); });
        report("row_copy", ms, 2048.0 * n, 0, "persistent row copy, 8 KiB in flight per wave");
        ms = timeit(reps, [&] { HIPCHECK(hipMemcpyAsync(dY, dX, (size_t)n * 1024, hipMemcpyDeviceToDevice, 0)); });
        report("hipMemcpyD2D", ms, 2048.0 * n, 0, "runtime copy");
    }

    // ---- product kernels through the C ABI
    {
        ndcn_csr a = {};
        a.n_rows = n; a.n_cols = n; a.nnz = (int64_t)nnz; a.rowptr = d_rp; a.colidx = d_ci; a.val = d_va;
        float ms = timeit(reps, [&] { ndcn_spmm_f32(&a, dX, nullptr, n, dY, 256, 1.f, 0, nullptr); });
        report("lib_wide", ms, alg, check(A, X, dY, "lib_wide"), "ndcn_spmm_f32 without a plan (spmm_wide_kernel)");
        // row-group union plan, 8 consecutive rows, cap 30 (the layout of csr.py:build_union_plan)
        const int R = 8, ng = (n + R - 1) / R;
        std::vector<int> ptr(ng + 1, 0), cols;
        std::vector<unsigned short> lidx(nnz + 32, 0);
        std::vector<int> u;
        for (int g = 0; g < ng; ++g) {
            u.clear();
            for (int r = g * R; r < std::min(n, g * R + R); ++r) for (int j = A.rp[r]; j < A.rp[r + 1]; ++j) u.push_back(A.ci[j]);
            std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
            for (int r = g * R; r < std::min(n, g * R + R); ++r) for (int j = A.rp[r]; j < A.rp[r + 1]; ++j)
                lidx[j] = (unsigned short)(std::lower_bound(u.begin(), u.end(), A.ci[j]) - u.begin());
            cols.insert(cols.end(), u.begin(), u.end());
            ptr[g + 1] = (int)cols.size();
        }
        int *d_ptr = dev(ptr), *d_cols = dev(cols);
        unsigned short *d_lidx = dev(lidx);
        a.ug_rows = R; a.ug_cap = 30; a.ug_ptr = d_ptr; a.ug_cols = d_cols; a.ug_lidx = d_lidx;
        ms = timeit(reps, [&] { ndcn_spmm_f32(&a, dX, nullptr, n, dY, 256, 1.f, 0, nullptr); });
        report("lib_union8", ms, alg, check(A, X, dY, "lib_union8"), "ndcn_spmm_f32 with the 8-row union plan (spmm_union_kernel)");
    }

    // ---- pipe variants
    std::vector<int> consec(n);
    for (int i = 0; i < n; ++i) consec[i] = i;
    auto patch_order = [&](int px, int py) {           // px x py lattice patches, patches walked along y then x
        std::vector<int> o; o.reserve(n);
        for (int bx = 0; bx < S; bx += px) for (int by = 0; by < S; by += py)
            for (int x = bx; x < std::min(S, bx + px); ++x) for (int y = by; y < std::min(S, by + py); ++y) o.push_back(x * S + y);
        return o;
    };
#define RUN_PIPE(NAME, CSR, XH, ORDER, R_, W_, CAPW_, NBUF_, DBG_, BPC, BYTES, NOTE)                                          \
    do {                                                                                                                       \
        PipePlan P;                                                                                                            \
        if (!build_plan(CSR, ORDER, R_, W_, CAPW_, P)) break;                                                                  \
        int *p_rows = dev(P.grp_rows), *p_uc = dev(P.ucols), *p_li = dev(P.lidx);                                              \
        int *p_rp = dev(CSR.rp);                                                                                               \
        float *p_va = dev(CSR.va, 32);                                                                                         \
        const size_t ldsb = (size_t)NBUF_ * CAPW_ * W_ * 1024;                                                                 \
        auto kern = spmm_pipe<R_, W_, CAPW_, NBUF_, DBG_>;                                                                     \
        HIPCHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));              \
        HIPCHECK(hipMemset(dY, 0xff, (size_t)n * 1024));                                                                       \
        float ms = timeit(reps, [&] { hipLaunchKernelGGL(kern, dim3(256 * BPC), dim3(64 * W_), ldsb, 0, p_rp, p_li, p_va, p_rows, p_uc, dX, dY, P.n_groups); }); \
        char note[256];                                                                                                        \
        snprintf(note, sizeof note, "%s; R %d W %d CAP %d NBUF %d lds %zu KiB, %d WG/CU, %.2f staged rows per row", NOTE, R_, W_, CAPW_ * W_, NBUF_, ldsb >> 10, BPC, P.loads_per_row); \
        report(NAME, ms, BYTES, DBG_ ? -1 : check(CSR, XH, dY, NAME), note);                                                   \
        (void)hipFree(p_rows); (void)hipFree(p_uc); (void)hipFree(p_li); (void)hipFree(p_rp); (void)hipFree(p_va);                                           \
    } while (0)

    const Csr I = identity(n);
    RUN_PIPE("pipe_identity_r8", I, X, consec, 8, 8, 1, 3, 0, 1, 2048.0 * n, "identity matrix through the pipe (copy ceiling)");
    RUN_PIPE("pipe_identity_r8_b4", I, X, consec, 8, 8, 1, 4, 0, 4, 2048.0 * n, "identity, 4 WG/CU");
    RUN_PIPE("pipe_identity_r16w16", I, X, consec, 16, 16, 1, 4, 0, 2, 2048.0 * n, "identity, 16 waves, 2 WG/CU");
    RUN_PIPE("pipe_r8", A, X, consec, 8, 8, 4, 3, 0, 1, alg, "8 consecutive rows");
    RUN_PIPE("pipe_r8_nb4", A, X, consec, 8, 8, 4, 4, 0, 1, alg, "8 consecutive rows, deeper ring");
    RUN_PIPE("pipe_r8_nb2_b2", A, X, consec, 8, 8, 4, 2, 0, 2, alg, "8 consecutive rows, 2 buffers, 2 WG/CU");
    RUN_PIPE("pipe_r8_nostore", A, X, consec, 8, 8, 4, 3, 1, 1, alg, "diagnostic: no stores");
    RUN_PIPE("pipe_r8_nodma", A, X, consec, 8, 8, 4, 3, 2, 1, alg, "diagnostic: no DMA");
    RUN_PIPE("pipe_r8_nosum", A, X, consec, 8, 8, 4, 3, 4, 1, alg, "diagnostic: no LDS reads / sums");
    RUN_PIPE("pipe_r16", A, X, consec, 16, 8, 7, 2, 0, 1, alg, "16 consecutive rows");
    RUN_PIPE("pipe_r16w16", A, X, consec, 16, 16, 4, 2, 0, 1, alg, "16 consecutive rows, 16 waves (cap 64)");
    {
        std::vector<int> o = patch_order(2, 4);
        RUN_PIPE("pipe_p2x4", A, X, o, 8, 8, 3, 3, 0, 2, alg, "2x4 lattice patches");
        RUN_PIPE("pipe_p2x4_nb4", A, X, o, 8, 8, 3, 4, 0, 1, alg, "2x4 lattice patches, deeper ring");
        o = patch_order(4, 4);
        RUN_PIPE("pipe_p4x4", A, X, o, 16, 8, 5, 3, 0, 1, alg, "4x4 lattice patches");
        RUN_PIPE("pipe_p4x4w16", A, X, o, 16, 16, 3, 3, 0, 1, alg, "4x4 lattice patches, 16 waves (cap 48)");
        o = patch_order(4, 8);
        RUN_PIPE("pipe_p4x8", A, X, o, 32, 16, 4, 2, 0, 1, alg, "4x8 lattice patches, 16 waves (cap 64)");
        o = patch_order(2, 8);
        RUN_PIPE("pipe_p2x8w16", A, X, o, 16, 16, 3, 3, 0, 1, alg, "2x8 lattice patches, 16 waves (cap 48)");
    }

#define RUN_REC(NAME, CSR, ORDER, R_, WC_, WD_, CAP_, D_, RECW_, DBG_, BPC, BYTES, NOTE)                                      \
    do {                                                                                                                       \
        RecPlan P;                                                                                                             \
        if (!build_rec(CSR, ORDER, R_, CAP_, RECW_, P)) break;                                                                 \
        int *p_rec = dev(P.rec, 1024);                                                                                         \
        const size_t ldsb = (size_t)((D_ + 1) * CAP_ + (2 * D_ + 1) * RECW_) * 1024;                                           \
        auto kern = spmm_rec<R_, WC_, WD_, CAP_, D_, RECW_, DBG_>;                                                             \
        HIPCHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));              \
        HIPCHECK(hipMemset(dY, 0xff, (size_t)n * 1024));                                                                       \
        float ms = timeit(reps, [&] { hipLaunchKernelGGL(kern, dim3(256 * BPC), dim3(64 * (WC_ + WD_)), ldsb, 0, p_rec, dX, dY, P.n_groups); }); \
        char note[256];                                                                                                        \
        snprintf(note, sizeof note, "%s; R %d WC %d WD %d CAP %d D %d RECW %d lds %zu KiB, %d WG/CU, %.2f staged rows per row", NOTE, R_, WC_, WD_, CAP_, D_, RECW_, ldsb >> 10, BPC, P.loads_per_row); \
        report(NAME, ms, BYTES, DBG_ ? -1 : check(CSR, X, dY, NAME), note);                                                    \
        (void)hipFree(p_rec);                                                                                                  \
    } while (0)

    RUN_REC("rec_identity", I, consec, 8, 8, 2, 8, 2, 1, 0, 1, 2048.0 * n, "identity (copy ceiling of this pipeline)");
    RUN_REC("rec_identity_b2", I, consec, 8, 8, 2, 8, 2, 1, 0, 2, 2048.0 * n, "identity, 2 WG/CU");
    RUN_REC("rec_identity_d4", I, consec, 8, 8, 2, 8, 4, 1, 0, 1, 2048.0 * n, "identity, depth 4");
    RUN_REC("rec_r8_d2_wd2", A, consec, 8, 8, 2, 32, 2, 1, 0, 1, alg, "8 consecutive rows");
    RUN_REC("rec_r8_d2_wd4", A, consec, 8, 8, 4, 32, 2, 1, 0, 1, alg, "8 consecutive rows");
    RUN_REC("rec_r8_d3_wd4", A, consec, 8, 8, 4, 32, 3, 1, 0, 1, alg, "8 consecutive rows");
    RUN_REC("rec_r8_d1_wd4_b2", A, consec, 8, 8, 4, 32, 1, 1, 0, 2, alg, "8 consecutive rows, 2 WG/CU");
    RUN_REC("rec_r8_d2_wd4_nostore", A, consec, 8, 8, 4, 32, 2, 1, 1, 1, alg, "diagnostic: no stores");
    RUN_REC("rec_r8_d2_wd4_nosum", A, consec, 8, 8, 4, 32, 2, 1, 4, 1, alg, "diagnostic: no LDS row reads / sums");
    RUN_REC("rec_r16_d1", A, consec, 16, 8, 4, 56, 1, 2, 0, 1, alg, "16 consecutive rows");
    {
        std::vector<int> o = patch_order(2, 4);
        RUN_REC("rec_p2x4_d2", A, o, 8, 8, 4, 24, 2, 1, 0, 1, alg, "2x4 patches");
        RUN_REC("rec_p2x4_d2_b2", A, o, 8, 8, 4, 24, 2, 1, 0, 2, alg, "2x4 patches, 2 WG/CU");
        RUN_REC("rec_p2x4_d3", A, o, 8, 8, 4, 24, 3, 1, 0, 1, alg, "2x4 patches");
        RUN_REC("rec_p2x4_d4", A, o, 8, 8, 4, 24, 4, 1, 0, 1, alg, "2x4 patches");
        o = patch_order(4, 4);
        RUN_REC("rec_p4x4_d2", A, o, 16, 8, 4, 40, 2, 2, 0, 1, alg, "4x4 patches");
        RUN_REC("rec_p4x4_d1_b2", A, o, 16, 8, 4, 36, 1, 2, 0, 2, alg, "4x4 patches, 2 WG/CU");
        o = patch_order(4, 8);
        RUN_REC("rec_p4x8_d1", A, o, 32, 8, 4, 60, 1, 3, 0, 1, alg, "4x8 patches");
    }
    return 0;
}
