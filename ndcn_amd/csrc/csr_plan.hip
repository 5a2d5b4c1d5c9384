// ndcn_csr_create / ndcn_csr_destroy: the operator handle of the C ABI (include/ndcn_hip.h).
//
// The reference holds its operator `A` by reference on the module (neural_dynamics.py:9-18) and the caller converts it
// once before model construction (heat_dynamics.py:170-175).  The counterpart here: the caller hands over a CSR it owns,
// once, and gets a handle whose view (struct ndcn_csr) carries the plans the H = 256 kernels select on - the group-record
// plan (spmm_rec.hip, rhs_fused3.hip), the lattice walk orders, the long-row plan (hub rows).  All of it is integer work
// on the CSR arrays and runs on the device, in the order of the arrays (no floating-point result depends on it):
//
//   stencil detection   one pass over the entries collects the distinct (column - row) offsets of the own columns in a
//                       64-slot table (a 2-D lattice in row-major order has <= 25); the stride is then a few host
//                       comparisons, the patch walk order a sort of the n / 16 patch ids
//   group records       one thread per group of 8 / 16 rows: the group's distinct columns by insertion into a sorted
//                       list of <= cap entries (a group that needs more is flagged and gathered directly by the
//                       kernels), then the rows' entries re-indexed into that list by binary search
//   long-row plan       degree histogram -> threshold; exclusive scans (hipcub) over the hub flags and the light
//                       operator's row lengths; one pass copies the hub rows' entries into their compact CSR over
//                       <= 256-entry segments and the light rows into the operator the fused kernel runs on
//   column-sweep plan   (operators without locality whose partial sums fit the register files: spmm_sweep.hip) the rows cut
//                       into slabs of <= 49 consecutive rows, one per wave; every entry packed as {row in slab << 24 | column,
//                       value}; a segmented radix sort (hipcub) by column inside each slab; slabs padded to groups of 8
//
// The decisions (which record shape, which threshold, whether a plan pays) are the ones ndcn_amd/csr.py took in Python
// up to ABI 10; tests/_plan_reference.py keeps that restatement and tests/test_gpu_plans.py compares bit for bit.
#include <algorithm>
#include <cmath>
#include <new>
#include <numeric>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "kernels.h"

struct ndcn_csr_handle {
    ndcn_csr v;
    std::vector<void *> owned;          // device allocations freed by ndcn_csr_destroy
    int32_t *group_order = nullptr;     // [n_group_order] row ids (-1: empty slot) in the order the records group them
    int64_t n_group_order = 0;
    int64_t stencil_stride = 0;
    int64_t staged_nnz = 0, staged_cols = 0;
    int32_t hub_threshold = 0;
    int64_t n_halo = 0;
    float *halo_S = nullptr;            // [n_halo + hub_n][H]: the kernels' second panel ([halo | hub rows])
    int H = 0;
    int64_t sweep_entries = 0;          // padded entries of the column-sweep plan
};

namespace ndcn {
namespace {

constexpr int kOffSlots = 64;            // offset table of the stencil detection (a 5 x 5 stencil has 25)
constexpr int kMaxCap = 48;              // largest column-list capacity of a record shape
constexpr int64_t kEmpty = INT64_MIN;

struct Shape { int rows, cap, kib; };
const Shape kShapes[3] = {{8, 32, 1}, {16, 40, 2}, {8, 48, 2}};      // spmm_rec.hip's variants

template <typename T>
int dev_alloc(ndcn_csr_handle *h, T **p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    void *q = nullptr;
    NDCN_HIP(hipMalloc(&q, count * sizeof(T)));
    h->owned.push_back(q);
    *p = static_cast<T *>(q);
    return NDCN_OK;
}

void dev_release(ndcn_csr_handle *h, void *p) {
    if (!p) return;
    auto it = std::find(h->owned.begin(), h->owned.end(), p);
    if (it != h->owned.end()) h->owned.erase(it);
    (void)hipFree(p);
}

bool env_off(const char *name) {
    const char *e = getenv(name);
    return e && e[0] == '0';
}

// ------------------------------------------------------------------------------------------------ stencil detection

// table[0..64): distinct offsets (kEmpty = free); table_n[0] = entries taken, table_n[1] = overflow flag
__global__ __launch_bounds__(256) void offsets_kernel(int64_t n, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colidx, int64_t row_base, int64_t n_own,
                                                      long long *table, int *table_n) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (*(volatile int *)(table_n + 1)) return;
    long long last = kEmpty;
    for (int32_t e = rowptr[r], end = rowptr[r + 1]; e < end; ++e) {
        const int64_t c = colidx[e];
        if (c >= n_own) continue;
        const long long off = c - (r + row_base);
        if (off == last) continue;
        last = off;
        unsigned slot = (unsigned)((unsigned long long)off * 0x9E3779B97F4A7C15ull >> 58);        // 6 bits
        for (int probe = 0; probe < kOffSlots; ++probe, slot = (slot + 1) & (kOffSlots - 1)) {
            long long cur = *(volatile long long *)(table + slot);
            if (cur == off) break;
            if (cur == kEmpty) {
                cur = (long long)atomicCAS((unsigned long long *)(table + slot), (unsigned long long)kEmpty, (unsigned long long)off);
                if (cur == kEmpty) { if (atomicAdd(table_n, 1) >= 48) atomicExch(table_n + 1, 1); break; }
                if (cur == off) break;
            }
            if (probe == kOffSlots - 1) atomicExch(table_n + 1, 1);
        }
        if (*(volatile int *)(table_n + 1)) return;
    }
}

// slots of the patch walk: patch t = perm[s / (px py)] of the PX x PY patch grid, slot (i, j) inside it
__global__ __launch_bounds__(256) void patch_expand_kernel(int64_t n_slots, const int32_t *__restrict__ perm, int px, int py,
                                                           int64_t PY, int64_t S, int64_t x_lo, int64_t x_hi, int64_t row_base,
                                                           int64_t n, int32_t *out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int per = px * py;
    const int64_t t = perm[s / per];
    const int q = (int)(s % per);
    const int64_t x = x_lo + (t / PY) * px + q / py, y = (t % PY) * py + q % py;
    const int64_t node = x * S + y;
    const bool ok = x < x_hi && y < S && node >= row_base && node < row_base + n;
    out[s] = ok ? (int32_t)(node - row_base) : -1;
}

// The stride of a 2-D lattice stencil in row-major node order from its offset set, or 0 (restated in tests/_plan_reference.py: detect_stencil_order).
int64_t stencil_stride(std::vector<long long> off) {
    if (off.empty() || off.size() > 25) return 0;
    // the smallest large |offset| is S - b_max with b_max <= 2 (a boundary band of a shard may see only the lattice row
    // above it among its own columns: magnitudes, not signs)
    long long bmin = 0;
    for (long long o : off) {
        const long long m = o < 0 ? -o : o;
        if (m > 2 && (bmin == 0 || m < bmin)) bmin = m;
    }
    if (bmin == 0) return 0;
    int64_t S = 0;
    double best_b = 3;
    for (long long cand = bmin; cand <= bmin + 2; ++cand) {
        bool ok = cand >= 8;
        double bmax = 0;
        for (long long o : off) {
            const double a = std::nearbyint((double)o / (double)cand);        // numpy rint: half to even
            const double b = (double)o - a * (double)cand;
            if (std::fabs(a) > 2) ok = false;
            bmax = std::max(bmax, std::fabs(b));
        }
        if (ok && bmax < best_b) { S = cand; best_b = (double)(int64_t)bmax; }
    }
    return S;
}

int detect_stencil(ndcn_csr_handle *h, int64_t row_base, int64_t n_own, bool hinted, hipStream_t st) {
    const ndcn_csr &A = h->v;
    const int64_t n = A.n_rows;
    if (A.nnz == 0 || n < 64 || (!hinted && A.n_rows != A.n_cols)) return NDCN_OK;
    long long *table;
    int *table_n;
    int rc;
    if ((rc = dev_alloc(h, &table, kOffSlots))) return rc;
    if ((rc = dev_alloc(h, &table_n, 2))) return rc;
    std::vector<long long> init(kOffSlots, kEmpty);
    NDCN_HIP(hipMemcpyAsync(table, init.data(), sizeof(long long) * kOffSlots, hipMemcpyHostToDevice, st));
    NDCN_HIP(hipMemsetAsync(table_n, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, A.rowptr, A.colidx, row_base, n_own,
                       table, table_n);
    NDCN_LAUNCH_CHECK();
    std::vector<long long> got(kOffSlots);
    int cnt[2];
    NDCN_HIP(hipMemcpyAsync(got.data(), table, sizeof(long long) * kOffSlots, hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipMemcpyAsync(cnt, table_n, sizeof(cnt), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    dev_release(h, table);
    dev_release(h, table_n);
    if (cnt[1]) return NDCN_OK;
    std::vector<long long> off;
    for (long long o : got)
        if (o != kEmpty) off.push_back(o);
    const int64_t S = stencil_stride(off);
    if (S == 0) return NDCN_OK;
    h->stencil_stride = S;
    // patches of px x py nodes in row-major patch order, each padded to px * py slots; inside every XCD range of the walk
    // (the group kernels give each of the 8 XCDs a contiguous range, its 32 workgroups take it round-robin) in strips
    // `strip` patches wide, top to bottom: what runs concurrently is one row of a strip, the next iteration the row below
    // (the two lattice rows they share are still in that XCD's L2)
    const int px = 4, py = 4, n_chunks = 8;
    const int64_t x_lo = row_base / S, x_hi = (row_base + n - 1) / S + 1;
    const int64_t PX = (x_hi - x_lo + px - 1) / px, PY = (S + py - 1) / py;
    const int64_t np = PX * PY;
    std::vector<int32_t> perm((size_t)np);
    std::iota(perm.begin(), perm.end(), 0);
    const char *se = getenv("NDCN_PATCH_STRIP");
    const int64_t strip = se ? atoll(se) : 32;
    if (strip > 0 && PX > 1 && PY > strip) {
        const int64_t per = (np + n_chunks - 1) / n_chunks;
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
            const int64_t Xa = a / PY, Ya = a % PY, Xb = b / PY, Yb = b % PY;
            if (a / per != b / per) return a / per < b / per;
            if (Ya / strip != Yb / strip) return Ya / strip < Yb / strip;
            if (Xa != Xb) return Xa < Xb;
            return Ya % strip < Yb % strip;
        });
    }
    int32_t *d_perm;
    if ((rc = dev_alloc(h, &d_perm, (size_t)np))) return rc;
    NDCN_HIP(hipMemcpyAsync(d_perm, perm.data(), sizeof(int32_t) * (size_t)np, hipMemcpyHostToDevice, st));
    const int64_t n_slots = np * px * py;
    if ((rc = dev_alloc(h, &h->group_order, (size_t)n_slots))) return rc;
    h->n_group_order = n_slots;
    hipLaunchKernelGGL(patch_expand_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, n_slots, d_perm, px, py, PY, S,
                       x_lo, x_hi, row_base, n, h->group_order);
    NDCN_LAUNCH_CHECK();
    NDCN_HIP(hipStreamSynchronize(st));          // perm leaves scope
    dev_release(h, d_perm);
    return NDCN_OK;
}

// Walk order of the fused RHS kernel's 64-row tiles on a lattice of stride S (tests/_plan_reference.py: lattice_tile_order): 32 consecutive
// positions - what an XCD runs concurrently - form a block of 32 lattice rows x 64 columns.
int lattice_tile_order(ndcn_csr_handle *h, int64_t S, hipStream_t st) {
    const int64_t n = h->v.n_rows, block_rows = 32, n_chunks = 8;
    const int64_t nt = (n + 63) / 64;
    const int64_t per = (nt + n_chunks - 1) / n_chunks;
    std::vector<int64_t> key((size_t)nt);
    for (int64_t t = 0; t < nt; ++t) {
        const int64_t x = (64 * t) / S, y = (64 * t) % S, band = t / per;
        key[(size_t)t] = ((band * (n / S / block_rows + 2) + x / block_rows) * (S / 64 + 2) + y / 64) * block_rows + x % block_rows;
    }
    std::vector<int32_t> order((size_t)nt);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[(size_t)a] < key[(size_t)b]; });
    int32_t *d;
    int rc = dev_alloc(h, &d, (size_t)nt);
    if (rc) return rc;
    NDCN_HIP(hipMemcpyAsync(d, order.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    NDCN_HIP(hipStreamSynchronize(st));
    h->v.tile_order = d;
    return NDCN_OK;
}

// ------------------------------------------------------------------------------------------------ group records

// One thread per group.  counters[0] += entries of the groups a record holds, counters[1] += their distinct columns.
__global__ __launch_bounds__(128) void rec_build_kernel(int64_t ng, int64_t M, const int32_t *__restrict__ order,
                                                        const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                        const float *__restrict__ val, int R, int CAP, int words, int32_t *rec,
                                                        unsigned long long *counters) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const int E0 = CAP + 2 * R, ecap = (words - E0) / 2;
    int32_t *base = rec + g * words;
    int32_t cols[kMaxCap];
    int nu = 0, gtot = 0, maxc = 0;
    for (int i = 0; i < R; ++i) {
        const int64_t p = g * R + i;
        const int32_t row = p < M ? (order ? order[p] : (int32_t)p) : -1;
        if (row < 0) continue;
        const int cnt = rowptr[row + 1] - rowptr[row];
        gtot += cnt;
        maxc = cnt > maxc ? cnt : maxc;
    }
    bool fits = gtot <= ecap && maxc <= 64;
    for (int i = 0; i < R && fits; ++i) {
        const int64_t p = g * R + i;
        const int32_t row = p < M ? (order ? order[p] : (int32_t)p) : -1;
        if (row < 0) continue;
        for (int32_t e = rowptr[row], end = rowptr[row + 1]; e < end; ++e) {
            const int32_t c = colidx[e];
            int lo = 0, hi = nu;                       // first position with cols[pos] >= c
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cols[mid] < c) lo = mid + 1; else hi = mid;
            }
            if (lo < nu && cols[lo] == c) continue;
            if (nu == CAP) { fits = false; break; }
            for (int k = nu; k > lo; --k) cols[k] = cols[k - 1];
            cols[lo] = c;
            ++nu;
        }
    }
    // column list: the distinct columns, padded with the last one (a record the group does not fit keeps zeros)
    if (fits)
        for (int k = 0; k < CAP; ++k) base[k] = nu ? cols[k < nu ? k : nu - 1] : 0;
    // headers {row id or -1, cnt | ofs << 16}; 0xffff flags a group the kernels gather directly
    int ofs = 0;
    for (int i = 0; i < R; ++i) {
        const int64_t p = g * R + i;
        const int32_t row = p < M ? (order ? order[p] : (int32_t)p) : -1;
        const int cnt = row < 0 ? 0 : rowptr[row + 1] - rowptr[row];
        base[CAP + 2 * i] = row;
        base[CAP + 2 * i + 1] = fits ? (cnt | (ofs << 16)) : 0xffff;
        if (fits && row >= 0) {
            const int32_t e0 = rowptr[row];
            for (int q = 0; q < cnt; ++q) {
                const int32_t c = colidx[e0 + q];
                int lo = 0, hi = nu;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cols[mid] < c) lo = mid + 1; else hi = mid;
                }
                base[E0 + 2 * (ofs + q)] = lo;
                base[E0 + 2 * (ofs + q) + 1] = __float_as_int(val[e0 + q]);
            }
        }
        ofs += cnt;
    }
    if (fits) {
        atomicAdd(counters, (unsigned long long)gtot);
        atomicAdd(counters + 1, (unsigned long long)nu);
    }
}

struct RecPlan { int32_t *rec = nullptr; Shape s{}; int64_t groups = 0, staged_nnz = 0, staged_cols = 0; double staged = 0, loads = 0; };

int build_rec(ndcn_csr_handle *h, const int32_t *order, int64_t M, Shape s, RecPlan *out, hipStream_t st) {
    const ndcn_csr &A = h->v;
    const int words = s.kib * 256;
    const int64_t ng = (M + s.rows - 1) / s.rows;
    int rc;
    int32_t *rec;
    unsigned long long *ctr;
    if ((rc = dev_alloc(h, &rec, (size_t)(ng > 0 ? ng : 1) * words))) return rc;
    if ((rc = dev_alloc(h, &ctr, 2))) return rc;
    NDCN_HIP(hipMemsetAsync(rec, 0, sizeof(int32_t) * (size_t)(ng > 0 ? ng : 1) * words, st));
    NDCN_HIP(hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned long long), st));
    if (ng > 0) {
        hipLaunchKernelGGL(rec_build_kernel, dim3((unsigned)((ng + 127) / 128)), dim3(128), 0, st, ng, M, order, A.rowptr, A.colidx,
                           A.val, s.rows, s.cap, words, rec, ctr);
        NDCN_LAUNCH_CHECK();
    }
    unsigned long long got[2];
    NDCN_HIP(hipMemcpyAsync(got, ctr, sizeof(got), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    dev_release(h, ctr);
    out->rec = rec;
    out->s = s;
    out->groups = ng;
    out->staged_nnz = (int64_t)got[0];
    out->staged_cols = (int64_t)got[1];
    out->staged = (double)out->staged_nnz / (double)std::max<int64_t>(A.nnz, 1);
    out->loads = (double)(out->staged_cols + (A.nnz - out->staged_nnz)) / (double)std::max<int64_t>(A.n_rows, 1);
    return NDCN_OK;
}

// ------------------------------------------------------------------------------------------------ long-row plan

__global__ __launch_bounds__(256) void degree_hist_kernel(int64_t n, const int32_t *__restrict__ rowptr, int t0, int t1, int t2,
                                                          unsigned long long *cnt) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d = r < n ? rowptr[r + 1] - rowptr[r] : 0;
    const unsigned long long m0 = __ballot(d > t0), m1 = __ballot(d > t1), m2 = __ballot(d > t2);
    if ((threadIdx.x & 63) == 0) {
        if (m0) atomicAdd(cnt, (unsigned long long)__popcll(m0));
        if (m1) atomicAdd(cnt + 1, (unsigned long long)__popcll(m1));
        if (m2) atomicAdd(cnt + 2, (unsigned long long)__popcll(m2));
    }
}

// flag[r] = row r is a hub; ltdeg[r] = its length in the light operator (a hub: one entry)
__global__ __launch_bounds__(256) void hub_flags_kernel(int64_t n, const int32_t *__restrict__ rowptr, int thr, int32_t *flag,
                                                        int32_t *ltdeg) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int d = rowptr[r + 1] - rowptr[r];
    flag[r] = d > thr;
    ltdeg[r] = d > thr ? 1 : d;
}

// hub h: its row id, its length and its number of <= seg-entry segments (hub_idx = exclusive scan of the flags)
__global__ __launch_bounds__(256) void hub_list_kernel(int64_t n, const int32_t *__restrict__ rowptr, int thr, int seg,
                                                       const int32_t *__restrict__ hub_idx, int32_t *hub_row, int32_t *hub_deg,
                                                       int32_t *hub_nseg) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int d = rowptr[r + 1] - rowptr[r];
    if (d > thr) {
        const int32_t hI = hub_idx[r];
        hub_row[hI] = (int32_t)r;
        hub_deg[hI] = d;
        hub_nseg[hI] = (d + seg - 1) / seg;
    }
}

// one workgroup per hub: its entries, compact, in stored order; its segments' extents; the combine operator's ones
__global__ __launch_bounds__(256) void hub_fill_kernel(int n_hub, int seg, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colidx, const float *__restrict__ val,
                                                       const int32_t *__restrict__ hub_row, const int32_t *__restrict__ hub_off,
                                                       const int32_t *__restrict__ cmb_rowptr, int32_t *hub_colidx, float *hub_val,
                                                       int32_t *seg_rowptr, int32_t *cmb_colidx, float *cmb_val) {
    const int hI = blockIdx.x;
    if (hI >= n_hub) return;
    const int32_t r = hub_row[hI], src = rowptr[r], dst = hub_off[hI], d = rowptr[r + 1] - src;
    for (int q = threadIdx.x; q < d; q += blockDim.x) {
        hub_colidx[dst + q] = colidx[src + q];
        hub_val[dst + q] = val[src + q];
    }
    const int32_t s0 = cmb_rowptr[hI], ns = (d + seg - 1) / seg;
    for (int k = threadIdx.x; k < ns; k += blockDim.x) {
        seg_rowptr[s0 + k] = dst + k * seg;
        cmb_colidx[s0 + k] = s0 + k;
        cmb_val[s0 + k] = 1.0f;
    }
}

// light operator: every hub row replaced by ONE entry (column n_cols + h, value 1)
__global__ __launch_bounds__(256) void light_fill_kernel(int64_t n, int64_t n_cols, int thr, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colidx, const float *__restrict__ val,
                                                         const int32_t *__restrict__ hub_idx, const int32_t *__restrict__ lt_rowptr,
                                                         int32_t *lt_colidx, float *lt_val) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t src = rowptr[r], d = rowptr[r + 1] - src, dst = lt_rowptr[r];
    if (d > thr) {
        lt_colidx[dst] = (int32_t)(n_cols + hub_idx[r]);
        lt_val[dst] = 1.0f;
    } else {
        for (int q = 0; q < d; ++q) {
            lt_colidx[dst + q] = colidx[src + q];
            lt_val[dst + q] = val[src + q];
        }
    }
}

__global__ void set_i32_kernel(int32_t *p, int32_t v) { *p = v; }

__global__ __launch_bounds__(256) void max_row_len_kernel(int64_t n, const int32_t *__restrict__ rowptr, int *out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int d = r < n ? rowptr[r + 1] - rowptr[r] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_down(d, off, 64));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, d);
}

// out[0..n] = exclusive prefix sums of in[0..n), out[n] = the total (int32: every sum here is bounded by nnz < 2^31)
int exclusive_scan(ndcn_csr_handle *h, const int32_t *in, int32_t *out, int64_t n, int32_t *h_total, hipStream_t st) {
    if (n == 0) {
        NDCN_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), st));
        *h_total = 0;
        return NDCN_OK;
    }
    size_t bytes = 0;
    NDCN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, st));
    char *tmp;
    int rc = dev_alloc(h, &tmp, bytes);
    if (rc) return rc;
    NDCN_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, (int)n, st));
    int32_t last_in = 0, last_out = 0;
    NDCN_HIP(hipMemcpyAsync(&last_in, in + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipMemcpyAsync(&last_out, out + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    *h_total = last_in + last_out;
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, st, out + n, *h_total);
    NDCN_LAUNCH_CHECK();
    dev_release(h, tmp);
    return NDCN_OK;
}

int build_hub_plan(ndcn_csr_handle *h, int H, int thr, bool external_scratch, hipStream_t st) {
    ndcn_csr &A = h->v;
    const int64_t n = A.n_rows;
    const int seg = 256;
    const unsigned gn = (unsigned)((n + 255) / 256);
    int rc;
    int32_t *flag, *ltdeg, *hub_idx, *lt_rowptr;
    if ((rc = dev_alloc(h, &flag, (size_t)n)) || (rc = dev_alloc(h, &ltdeg, (size_t)n)) || (rc = dev_alloc(h, &hub_idx, (size_t)n + 1)) ||
        (rc = dev_alloc(h, &lt_rowptr, (size_t)n + 1)))
        return rc;
    hipLaunchKernelGGL(hub_flags_kernel, dim3(gn), dim3(256), 0, st, n, A.rowptr, thr, flag, ltdeg);
    NDCN_LAUNCH_CHECK();
    int32_t n_hub = 0, lt_nnz = 0, hub_nnz = 0, nseg = 0;
    if ((rc = exclusive_scan(h, flag, hub_idx, n, &n_hub, st))) return rc;
    if ((rc = exclusive_scan(h, ltdeg, lt_rowptr, n, &lt_nnz, st))) return rc;
    dev_release(h, flag);
    dev_release(h, ltdeg);
    if (n_hub == 0) {
        dev_release(h, hub_idx);
        dev_release(h, lt_rowptr);
        return NDCN_OK;
    }
    int32_t *hub_row, *hub_deg, *hub_nsegs, *hub_off, *cmb_rowptr;
    if ((rc = dev_alloc(h, &hub_row, (size_t)n_hub)) || (rc = dev_alloc(h, &hub_deg, (size_t)n_hub)) ||
        (rc = dev_alloc(h, &hub_nsegs, (size_t)n_hub)) || (rc = dev_alloc(h, &hub_off, (size_t)n_hub + 1)) ||
        (rc = dev_alloc(h, &cmb_rowptr, (size_t)n_hub + 1)))
        return rc;
    hipLaunchKernelGGL(hub_list_kernel, dim3(gn), dim3(256), 0, st, n, A.rowptr, thr, seg, hub_idx, hub_row, hub_deg, hub_nsegs);
    NDCN_LAUNCH_CHECK();
    if ((rc = exclusive_scan(h, hub_deg, hub_off, n_hub, &hub_nnz, st))) return rc;
    if ((rc = exclusive_scan(h, hub_nsegs, cmb_rowptr, n_hub, &nseg, st))) return rc;
    dev_release(h, hub_deg);
    dev_release(h, hub_nsegs);
    int32_t *hub_colidx, *seg_rowptr, *cmb_colidx, *lt_colidx;
    float *hub_val, *cmb_val, *lt_val;
    if ((rc = dev_alloc(h, &hub_colidx, (size_t)hub_nnz)) || (rc = dev_alloc(h, &hub_val, (size_t)hub_nnz)) ||
        (rc = dev_alloc(h, &seg_rowptr, (size_t)nseg + 1)) || (rc = dev_alloc(h, &cmb_colidx, (size_t)nseg)) ||
        (rc = dev_alloc(h, &cmb_val, (size_t)nseg)) || (rc = dev_alloc(h, &lt_colidx, (size_t)lt_nnz)) ||
        (rc = dev_alloc(h, &lt_val, (size_t)lt_nnz)))
        return rc;
    hipLaunchKernelGGL(hub_fill_kernel, dim3((unsigned)n_hub), dim3(256), 0, st, n_hub, seg, A.rowptr, A.colidx, A.val, hub_row, hub_off,
                       cmb_rowptr, hub_colidx, hub_val, seg_rowptr, cmb_colidx, cmb_val);
    NDCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, st, seg_rowptr + nseg, hub_nnz);
    NDCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(light_fill_kernel, dim3(gn), dim3(256), 0, st, n, A.n_cols, thr, A.rowptr, A.colidx, A.val, hub_idx, lt_rowptr,
                       lt_colidx, lt_val);
    NDCN_LAUNCH_CHECK();
    NDCN_HIP(hipStreamSynchronize(st));
    dev_release(h, hub_idx);
    dev_release(h, hub_row);
    dev_release(h, hub_off);
    A.hub_n = n_hub;
    A.hub_nseg = nseg;
    A.hub_H = H;
    A.hub_nnz = hub_nnz;
    A.lt_nnz = lt_nnz;
    A.hub_seg_rowptr = seg_rowptr;
    A.hub_colidx = hub_colidx;
    A.hub_val = hub_val;
    A.hub_cmb_rowptr = cmb_rowptr;
    A.hub_cmb_colidx = cmb_colidx;
    A.hub_cmb_val = cmb_val;
    A.lt_rowptr = lt_rowptr;
    A.lt_colidx = lt_colidx;
    A.lt_val = lt_val;
    h->hub_threshold = thr;
    if (!external_scratch) {
        // [halo rows | hub rows] in ONE buffer: a shard's exchange receives into its head, the hubs' (A X) rows are its tail
        float *Sseg, *halo_S;
        if ((rc = dev_alloc(h, &Sseg, (size_t)nseg * H)) || (rc = dev_alloc(h, &halo_S, (size_t)(h->n_halo + n_hub) * H))) return rc;
        A.hub_Sseg = Sseg;
        h->halo_S = halo_S;
        A.hub_S = halo_S + h->n_halo * (int64_t)H;
    }
    return NDCN_OK;
}


// ---- column-sweep plan (struct ndcn_csr: sweep_*; the kernel: spmm_sweep.hip) ---------------------------------------------------
// Geometry, shared with spmm_sweep_f32: pass p holds rows [p R, min(n, (p + 1) R)), R = sweep_rows_per_pass; its np rows are cut
// into 8 XCD chunks of ceil(np / 8) rows, every chunk into 256 slabs of ceil(chunk / 256) <= 49 consecutive rows.
constexpr int kSweepRows = 49, kSweepSlots = 256, kSweepSlabs = kXcds * kSweepSlots;

struct SweepGeom { int64_t n, rpp; };
__device__ __host__ inline void sweep_slab_rows(const SweepGeom g, int64_t s, int64_t *row0, int64_t *row1) {
    const int64_t p = s / kSweepSlabs, x = (s % kSweepSlabs) / kSweepSlots, sl = s % kSweepSlots;
    const int64_t base = p * g.rpp, end = (p + 1) * g.rpp < g.n ? (p + 1) * g.rpp : g.n, np = end - base;
    const int64_t per_xcd = (np + kXcds - 1) / kXcds, rpw = (per_xcd + kSweepSlots - 1) / kSweepSlots;
    const int64_t xend = base + (x + 1) * per_xcd < end ? base + (x + 1) * per_xcd : end;
    int64_t r0 = base + x * per_xcd + sl * rpw, r1 = r0 + rpw;
    if (r0 > xend) r0 = xend;
    if (r1 > xend) r1 = xend;
    *row0 = r0, *row1 = r1;
}

// per slab: where its entries start in the CSR arrays (its rows are consecutive), how many there are, the count padded to 8
__global__ __launch_bounds__(256) void sweep_slab_kernel(int64_t nslab, SweepGeom g, const int32_t *__restrict__ rowptr, int32_t *seg_off,
                                                         int32_t *cnt_pad) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nslab) return;
    if (s == nslab) { seg_off[s] = rowptr[g.n]; return; }
    int64_t r0, r1;
    sweep_slab_rows(g, s, &r0, &r1);
    seg_off[s] = rowptr[r0];
    cnt_pad[s] = (rowptr[r1] - rowptr[r0] + 7) & ~7;
}

// per entry: {row within its slab << 24 | column, value bits} as ONE 64-bit word (low word first in memory)
__global__ __launch_bounds__(256) void sweep_pack_kernel(SweepGeom g, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                         const float *__restrict__ val, unsigned long long *packed) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= g.n) return;
    const int64_t p = r / g.rpp, base = p * g.rpp, end = (p + 1) * g.rpp < g.n ? (p + 1) * g.rpp : g.n, np = end - base;
    const int64_t per_xcd = (np + kXcds - 1) / kXcds, rpw = (per_xcd + kSweepSlots - 1) / kSweepSlots;
    const unsigned local = (unsigned)(((r - base) % per_xcd) % rpw);
    for (int32_t j = rowptr[r]; j < rowptr[r + 1]; ++j)
        packed[j] = ((unsigned long long)__float_as_uint(val[j]) << 32) | (unsigned long long)(local << 24 | (unsigned)colidx[j]);
}

__global__ __launch_bounds__(256) void sweep_pad_kernel(int64_t n_words, unsigned long long *ent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) ent[i] = (unsigned long long)((unsigned)kSweepRows << 24);      // 0 * X[0] into the dummy row
}

// one workgroup per slab: its sorted entries into its padded place, and the slab's table row
__global__ __launch_bounds__(256) void sweep_place_kernel(const int32_t *__restrict__ seg_off, const int32_t *__restrict__ pad_start,
                                                          const unsigned long long *__restrict__ sorted, unsigned long long *ent,
                                                          int32_t *slab) {
    const int64_t s = blockIdx.x;
    const int32_t src = seg_off[s], cnt = seg_off[s + 1] - src, dst = pad_start[s];
    for (int32_t i = threadIdx.x; i < cnt; i += blockDim.x) ent[dst + i] = sorted[src + i];
    if (threadIdx.x == 0) { slab[2 * s] = dst; slab[2 * s + 1] = cnt; }
}

__global__ __launch_bounds__(256) void sweep_eye_kernel(int64_t n, int32_t *rowptr, int32_t *colidx, float *val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) rowptr[i] = (int32_t)i;
    if (i < n) { colidx[i] = (int32_t)i; val[i] = 1.0f; }
}

// Worth it when the sweep moves clearly fewer rows of X through the fabric than the gather: 8 XCDs x n_cols rows per pass
// against nnz.  Measured on G(n,p) graphs of mean degree 40 (tools/micro/sweep_sizes.py, profiles/r04i_sweep_sizes.txt):
// sweep = 3.8e-8 ms per entry + 7.8e-8 ms per row an XCD pulls in, row gather = 1.44e-7 ms per entry - break-even at
// nnz = 0.73 x (8 passes n_cols); n = 10^5: 0.22 against 0.58 ms, 3 x 10^5 (3 passes): 1.06 against 1.78, 5 x 10^5: 2.19 against 2.94.
// The right-hand side pays one more write and read of S for the sweep, so the plan is taken from 1.5 x upwards, and only
// when the panel is larger than what an XCD's L2 holds anyway (8 192 rows of 1 KiB = 2 x 4 MiB).
bool sweep_pays(const ndcn_csr &A, int passes) {
    return device_is_whole_chip() && A.n_cols >= 8192 && (double)A.nnz >= 1.5 * (double)passes * kXcds * (double)A.n_cols;   // (the sweep needs the whole chip: common.h)
}

int build_sweep_plan(ndcn_csr_handle *h, bool external_scratch, hipStream_t st) {
    ndcn_csr &A = h->v;
    const int64_t n = A.n_rows, cap = (int64_t)kSweepSlabs * kSweepRows;        // 100 352 rows per pass
    const int passes = (int)((n + cap - 1) / cap);
    const int64_t rpp = (n + passes - 1) / passes, nslab = (int64_t)passes * kSweepSlabs;
    const SweepGeom g{n, rpp};
    int rc;
    int32_t *seg_off, *cnt_pad, *pad_start, *slab;
    unsigned long long *packed, *sorted, *ent;
    int32_t *keys_out;
    if ((rc = dev_alloc(h, &seg_off, (size_t)nslab + 1)) || (rc = dev_alloc(h, &cnt_pad, (size_t)nslab)) ||
        (rc = dev_alloc(h, &pad_start, (size_t)nslab + 1)) || (rc = dev_alloc(h, &slab, (size_t)nslab * 2)) ||
        (rc = dev_alloc(h, &packed, (size_t)A.nnz)) || (rc = dev_alloc(h, &sorted, (size_t)A.nnz)) || (rc = dev_alloc(h, &keys_out, (size_t)A.nnz)))
        return rc;
    hipLaunchKernelGGL(sweep_slab_kernel, dim3((unsigned)((nslab + 256) / 256)), dim3(256), 0, st, nslab, g, A.rowptr, seg_off, cnt_pad);
    NDCN_LAUNCH_CHECK();
    int32_t total = 0;
    if ((rc = exclusive_scan(h, cnt_pad, pad_start, nslab, &total, st))) return rc;
    hipLaunchKernelGGL(sweep_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, A.rowptr, A.colidx, A.val, packed);
    NDCN_LAUNCH_CHECK();
    // the slab's entries by column (ties - two rows of a slab referencing one column - fold into different accumulators)
    int end_bit = 1;
    while (end_bit < 31 && (1ll << end_bit) < A.n_cols) ++end_bit;
    size_t bytes = 0;
    NDCN_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, reinterpret_cast<const uint32_t *>(A.colidx),
                                                        reinterpret_cast<uint32_t *>(keys_out), packed, sorted, (int)A.nnz, (int)nslab, seg_off,
                                                        seg_off + 1, 0, end_bit, st));
    char *tmp;
    if ((rc = dev_alloc(h, &tmp, bytes ? bytes : 16))) return rc;
    NDCN_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, bytes, reinterpret_cast<const uint32_t *>(A.colidx), reinterpret_cast<uint32_t *>(keys_out),
                                                        packed, sorted, (int)A.nnz, (int)nslab, seg_off, seg_off + 1, 0, end_bit, st));
    const int64_t n_words = (int64_t)total + 32;                 // + 4 groups: the kernel prefetches two groups past a slab's end
    if ((rc = dev_alloc(h, &ent, (size_t)n_words))) return rc;
    hipLaunchKernelGGL(sweep_pad_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, n_words, ent);
    NDCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sweep_place_kernel, dim3((unsigned)nslab), dim3(256), 0, st, seg_off, pad_start, sorted, ent, slab);
    NDCN_LAUNCH_CHECK();
    uint32_t *prog;
    int32_t *eye_rowptr, *eye_colidx;
    float *eye_val;
    if ((rc = dev_alloc(h, &prog, (size_t)nslab + passes)) || (rc = dev_alloc(h, &eye_rowptr, (size_t)n + 1)) || (rc = dev_alloc(h, &eye_colidx, (size_t)n)) ||
        (rc = dev_alloc(h, &eye_val, (size_t)n)))
        return rc;
    NDCN_HIP(hipMemsetAsync(prog, 0, ((size_t)nslab + passes) * sizeof(uint32_t), st));     // progress words + one launch counter per pass
    hipLaunchKernelGGL(sweep_eye_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, n, eye_rowptr, eye_colidx, eye_val);
    NDCN_LAUNCH_CHECK();
    NDCN_HIP(hipStreamSynchronize(st));
    // the identity operator's 16-row group records: the dense stage of the right-hand side runs rhs_fused3 over S
    {
        ndcn_csr_handle eye;
        memset(&eye.v, 0, sizeof(eye.v));
        eye.v.n_rows = eye.v.n_cols = eye.v.nnz = n;
        eye.v.rowptr = eye_rowptr, eye.v.colidx = eye_colidx, eye.v.val = eye_val;
        RecPlan rp;
        rc = build_rec(&eye, nullptr, n, kShapes[1], &rp, st);
        for (void *q : eye.owned) h->owned.push_back(q);
        eye.owned.clear();
        if (rc) return rc;
        A.sweep_eye_rec = rp.rec;
        A.sweep_eye_groups = (int32_t)rp.groups;
    }
    dev_release(h, seg_off);
    dev_release(h, cnt_pad);
    dev_release(h, pad_start);
    dev_release(h, packed);
    dev_release(h, sorted);
    dev_release(h, keys_out);
    dev_release(h, tmp);
    const int64_t per_xcd = (rpp + kXcds - 1) / kXcds;
    A.sweep_passes = passes;
    A.sweep_rpw = (int32_t)((per_xcd + kSweepSlots - 1) / kSweepSlots);
    A.sweep_rows_per_pass = rpp;
    // column blocks of 2048 rows of X (2 MiB) and a window of 2: every wave of an XCD within two blocks of the slowest
    // (tools/micro/sweep_lab.hip: 0.196 ms; 1024 x 3: 0.200; 512 x 6: 0.227; no synchronisation: 0.341 - and on BASELINE config 2's RK4
    // step, alternating runs in one box: 1.258 ms against 1.279 for 1024 x 3, 1.262 for 1024 x 4, 1.38 for 1024 x 2)
    A.sweep_logb = 11;
    A.sweep_window = 2;
    A.sweep_ent = reinterpret_cast<const uint32_t *>(ent);
    A.sweep_slab = slab;
    A.sweep_prog = prog;
    A.sweep_eye_rowptr = eye_rowptr;
    A.sweep_eye_colidx = eye_colidx;
    A.sweep_eye_val = eye_val;
    h->sweep_entries = total;
    if (!external_scratch) {
        float *S;
        if ((rc = dev_alloc(h, &S, (size_t)n * 256))) return rc;
        A.sweep_S = S;
    }
    return NDCN_OK;
}

int choose_hub_threshold(ndcn_csr_handle *h, int forced, int *thr_out, hipStream_t st) {
    // Rows longer than the threshold leave the fused kernel.  Worth it only when such rows are the exception (measured,
    // 10^6 nodes: Barabasi-Albert m = 5 36.8 -> 24.5 ms/step at threshold 32; G(n,p) with mean degree 41, where 32 moves
    // nearly every row, 53 -> 57): the lowest of 32 / 64 / 128 that moves at most 5 % of the rows.
    *thr_out = 0;
    const ndcn_csr &A = h->v;
    if (A.nnz == 0 || A.n_rows == 0) return NDCN_OK;
    const char *env = getenv("NDCN_HUB_THRESHOLD");
    if (forced > 0) { *thr_out = forced; return NDCN_OK; }
    if (env) { *thr_out = atoi(env) > 0 ? atoi(env) : 0; return NDCN_OK; }
    unsigned long long *cnt;
    int rc = dev_alloc(h, &cnt, 3);
    if (rc) return rc;
    NDCN_HIP(hipMemsetAsync(cnt, 0, 3 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(degree_hist_kernel, dim3((unsigned)((A.n_rows + 255) / 256)), dim3(256), 0, st, A.n_rows, A.rowptr, 32, 64, 128, cnt);
    NDCN_LAUNCH_CHECK();
    unsigned long long got[3];
    NDCN_HIP(hipMemcpyAsync(got, cnt, sizeof(got), hipMemcpyDeviceToHost, st));
    NDCN_HIP(hipStreamSynchronize(st));
    dev_release(h, cnt);
    const int thrs[3] = {32, 64, 128};
    for (int i = 0; i < 3; ++i) {
        if (got[i] == 0) break;
        if ((double)got[i] <= 0.05 * (double)A.n_rows) { *thr_out = thrs[i]; break; }
    }
    return NDCN_OK;
}

int csr_create_plans(ndcn_csr_handle *h, int H, const ndcn_csr_hints *hints, hipStream_t st) {
    ndcn_csr &A = h->v;
    ndcn_csr_hints none;
    memset(&none, 0, sizeof(none));
    const ndcn_csr_hints &hi = hints ? *hints : none;
    h->H = H;
    h->n_halo = hi.n_halo > 0 ? hi.n_halo : 0;
    A.row_order = hi.row_order;
    if (A.n_rows > 0 && A.nnz > 0) {                        // the longest row (ndcn_csr::max_row_len)
        int *d_max;
        int rc0 = dev_alloc(h, &d_max, 1);
        if (rc0) return rc0;
        NDCN_HIP(hipMemsetAsync(d_max, 0, sizeof(int), st));
        hipLaunchKernelGGL(max_row_len_kernel, dim3((unsigned)((A.n_rows + 255) / 256)), dim3(256), 0, st, A.n_rows, A.rowptr, d_max);
        NDCN_LAUNCH_CHECK();
        int got = 0;
        NDCN_HIP(hipMemcpyAsync(&got, d_max, sizeof(int), hipMemcpyDeviceToHost, st));
        NDCN_HIP(hipStreamSynchronize(st));
        dev_release(h, d_max);
        A.max_row_len = got;
    }
    if (H != 256) return NDCN_OK;                       // the plans serve the H = 256 kernels
    int rc;
    // ---- long-row plan
    if (!(hi.flags & NDCN_PLAN_NO_HUB) && hi.hub_threshold >= 0) {
        int thr = 0;
        if ((rc = choose_hub_threshold(h, hi.hub_threshold, &thr, st))) return rc;
        if (thr > 0 && (rc = build_hub_plan(h, H, thr, hi.flags & NDCN_PLAN_EXTERNAL_SCRATCH, st))) return rc;
    }
    const bool order_only = hi.flags & NDCN_PLAN_ORDER_ONLY;
    if (A.nnz == 0 || (!order_only && ((hi.flags & NDCN_PLAN_NO_REC) || env_off("NDCN_REC_PLAN")))) return NDCN_OK;
    // ---- walk order: the caller's, or the patches of a detected lattice
    const int32_t *order = nullptr;
    int64_t M = A.n_rows;
    bool hinted = false;
    const bool lattice_hint = hi.lattice_n_own > 0;
    if (hi.group_order) {
        order = hi.group_order, M = hi.n_group_order, hinted = true;
    } else if (hi.row_order) {
        order = hi.row_order, hinted = true;
    } else if (!(hi.flags & NDCN_PLAN_NO_STENCIL) && !env_off("NDCN_REC_STENCIL")) {
        if ((rc = detect_stencil(h, lattice_hint ? hi.lattice_row_base : 0, lattice_hint ? hi.lattice_n_own : A.n_cols, lattice_hint, st)))
            return rc;
        if (h->group_order) {
            order = h->group_order, M = h->n_group_order, hinted = true;
            if (!(hi.flags & NDCN_PLAN_NO_TILE_ORDER) && !env_off("NDCN_TILE_ORDER") && !lattice_hint &&
                (rc = lattice_tile_order(h, h->stencil_stride, st)))
                return rc;
        }
    }
    if (order_only) return NDCN_OK;
    // ---- record shapes: with a walk order the lattice shapes; without, 8 consecutive rows with a 32-column list, or -
    // when that covers too few groups (ring neighbours + random shortcuts reference ~28 distinct columns) - 48.  A shape is
    // kept when it covers the operator and stages clearly fewer rows than a direct gather fetches.
    RecPlan best;
    if (hi.rec_rows > 0) {
        Shape s{hi.rec_rows, hi.rec_cap, hi.rec_kib};
        bool known = false;
        for (const Shape &k : kShapes) known |= k.rows == s.rows && k.cap == s.cap && k.kib == s.kib;
        if (!known) { set_error("ndcn_csr_create: record shape {%d, %d, %d} has no kernel", s.rows, s.cap, s.kib); return NDCN_EINVAL; }
        if ((rc = build_rec(h, order, M, s, &best, st))) return rc;
    } else {
        const double avg = (double)A.nnz / (double)std::max<int64_t>(A.n_rows, 1);
        const Shape cand[2] = {hinted ? kShapes[1] : kShapes[0], hinted ? kShapes[0] : kShapes[2]};
        for (const Shape &s : cand) {
            RecPlan p;
            if ((rc = build_rec(h, order, M, s, &p, st))) return rc;
            if (p.staged >= 0.9 && p.loads <= 0.75 * avg && (!best.rec || p.loads < best.loads)) {
                dev_release(h, best.rec);
                best = p;
            } else {
                dev_release(h, p.rec);
            }
        }
    }
    if (best.rec) {
        A.rec_rows = best.s.rows, A.rec_cap = best.s.cap, A.rec_kib = best.s.kib;
        A.rec_groups = (int32_t)best.groups;
        A.rec = best.rec;
        h->staged_nnz = best.staged_nnz, h->staged_cols = best.staged_cols;
    }
    return NDCN_OK;
}

int csr_create(ndcn_csr_handle *h, int H, const ndcn_csr_hints *hints, hipStream_t st) {
    int rc = csr_create_plans(h, H, hints, st);
    if (rc) return rc;
    // ---- column-sweep plan: whole operators without a group-record plan whose rows are long relative to their count
    ndcn_csr &A = h->v;
    const uint32_t flags = hints ? hints->flags : 0;
    const bool whole = !hints || (hints->lattice_n_own == 0 && hints->n_halo == 0);
    if (H != 256 || A.nnz == 0 || A.n_rows == 0 || !whole || A.rec || (flags & (NDCN_PLAN_NO_SWEEP | NDCN_PLAN_ORDER_ONLY)) || env_off("NDCN_SWEEP_PLAN"))
        return NDCN_OK;
    if (A.n_cols * (int64_t)1024 >= (1ll << 32) || A.n_cols >= (1 << 24)) return NDCN_OK;
    const int passes = (int)((A.n_rows + (int64_t)kSweepSlabs * kSweepRows - 1) / ((int64_t)kSweepSlabs * kSweepRows));
    if (!(flags & NDCN_PLAN_FORCE_SWEEP) && !(passes <= 4 && sweep_pays(A, passes))) return NDCN_OK;
    return build_sweep_plan(h, flags & NDCN_PLAN_EXTERNAL_SCRATCH, st);
}

}  // namespace
}  // namespace ndcn

using namespace ndcn;

extern "C" {

int ndcn_csr_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr, const int32_t *colidx, const float *val,
                    int H, const ndcn_csr_hints *hints, void *stream, ndcn_csr_handle **out) {
    NDCN_CHECK_ARG(out, "null output");
    *out = nullptr;
    NDCN_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < (1ll << 31) - 1 && nnz < (1ll << 31) - 1 && n_cols < (1ll << 31) - 1,
                   "operator dimensions out of range");
    NDCN_CHECK_ARG(rowptr && (nnz == 0 || (colidx && val)), "null CSR array");
    NDCN_CHECK_ARG(H > 0, "H must be positive");
    NDCN_CHECK_ARG(!hints || (hints->lattice_n_own >= 0 && hints->lattice_n_own <= n_cols && hints->lattice_row_base >= 0),
                   "lattice hint out of range");
    NDCN_CHECK_ARG(!hints || !hints->group_order || hints->n_group_order >= n_rows, "group order shorter than the row count");
    ndcn_csr_handle *h = new (std::nothrow) ndcn_csr_handle();
    if (!h) { set_error("out of host memory"); return NDCN_EINVAL; }
    memset(&h->v, 0, sizeof(h->v));
    h->v.n_rows = n_rows, h->v.n_cols = n_cols, h->v.nnz = nnz;
    h->v.rowptr = rowptr, h->v.colidx = colidx, h->v.val = val;
    const int rc = csr_create(h, H, hints, static_cast<hipStream_t>(stream));
    if (rc) { ndcn_csr_destroy(h); return rc; }
    *out = h;
    return NDCN_OK;
}

const ndcn_csr *ndcn_csr_view(const ndcn_csr_handle *h) { return h ? &h->v : nullptr; }

int ndcn_csr_info(const ndcn_csr_handle *h, int64_t o[16]) {
    NDCN_CHECK_ARG(h && o, "null argument");
    const ndcn_csr &A = h->v;
    const int64_t v[16] = {A.rec ? A.rec_rows : 0, A.rec ? A.rec_cap : 0, A.rec ? A.rec_kib : 0, A.rec ? A.rec_groups : 0,
                           h->staged_nnz, h->staged_cols, h->stencil_stride, h->n_group_order,
                           A.hub_n, A.hub_nseg, h->hub_threshold, A.hub_nnz, A.lt_nnz, h->n_halo, A.tile_order ? 1 : 0, h->H};
    memcpy(o, v, sizeof(v));
    return NDCN_OK;
}

const int32_t *ndcn_csr_group_order(const ndcn_csr_handle *h) { return h ? h->group_order : nullptr; }

float *ndcn_csr_halo_panel(const ndcn_csr_handle *h) { return h ? h->halo_S : nullptr; }

int ndcn_csr_set_hub_scratch(ndcn_csr_handle *h, float *Sseg, float *halo_S) {
    NDCN_CHECK_ARG(h, "null handle");
    if (h->v.hub_n == 0) return NDCN_OK;
    NDCN_CHECK_ARG(Sseg && halo_S && aligned16(Sseg) && aligned16(halo_S), "hub scratch must be two 16-byte aligned device buffers");
    h->v.hub_Sseg = Sseg;
    h->halo_S = halo_S;
    h->v.hub_S = halo_S + h->n_halo * (int64_t)h->v.hub_H;
    return NDCN_OK;
}

int ndcn_csr_sweep_info(const ndcn_csr_handle *h, int64_t o[8]) {
    NDCN_CHECK_ARG(h && o, "null argument");
    const ndcn_csr &A = h->v;
    const int64_t v[8] = {A.sweep_ent ? A.sweep_passes : 0, A.sweep_ent ? A.sweep_rpw : 0, A.sweep_ent ? A.sweep_logb : 0,
                          A.sweep_ent ? A.sweep_window : 0, h->sweep_entries, A.sweep_ent ? A.sweep_rows_per_pass : 0, A.sweep_S ? 1 : 0, 0};
    memcpy(o, v, sizeof(v));
    return NDCN_OK;
}

int ndcn_csr_set_sweep_scratch(ndcn_csr_handle *h, float *S) {
    NDCN_CHECK_ARG(h, "null handle");
    if (!h->v.sweep_ent) return NDCN_OK;
    NDCN_CHECK_ARG(S && aligned16(S), "the sweep scratch must be a 16-byte aligned device buffer of n_rows x 256 floats");
    h->v.sweep_S = S;
    return NDCN_OK;
}

int ndcn_csr_destroy(ndcn_csr_handle *h) {
    if (!h) return NDCN_OK;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return NDCN_OK;
}

}  // extern "C"
