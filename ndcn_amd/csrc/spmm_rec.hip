// Group-record SpMM (H = 256):  Y = alpha (A X) [relu]  - and, for the `no_control` right-hand side relu(A X)
// (neural_dynamics.py:29,32,36; the README dgnn command), the Runge-Kutta algebra that consumes it in the same pass
// (COMBINE / ERROR / RK4 as in rhs_fused2.hip; rk_common.py:51,60,72-78; misc.py:146-157).
//
// Why this kernel exists (measured on MI355X, 1M-node lattice, tools/micro/spmm_lab.hip): a gather that fetches every
// neighbour row through the vector-memory pipe into registers is bound by request latency chains, not by HBM - the
// row SpMM ran at 0.76 ms, the union-in-LDS kernel of round 1 at 0.59 ms, a plain row copy at 0.44 ms.  Here EVERY
// byte a group of rows needs arrives by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) at addresses that
// are known without loading streamed index data first, several groups ahead of its use:
//
//   plan (ndcn_csr::rec, built once per operator by ndcn_csr_create: csr_plan.hip): rows are cut into groups of R
//   (consecutive in the operator's walk order); group g owns one fixed-size RECORD of RECW KiB
//       words [0, CAP)            the DISTINCT columns the group's rows reference (its "union"), padded by repetition
//       words [CAP, CAP + 2R)     per row {row id | -1, cnt | ofs << 16}   (cnt = 0xffff: group does not fit, see below)
//       words [CAP + 2R, ...)     the rows' entries as (slot in the union, value) pairs, row after row
//
//   one persistent workgroup per CU = 4 DMA waves + 8 compute waves, walking the groups of its XCD's chunk:
//     DMA waves      iteration it: wait (counted vmcnt) for their share of group it; barrier; issue the record of
//                    group it + 2D and the union rows of group it + D, whose column ids they read out of the record
//                    that landed D iterations earlier.  They never touch a register with data.
//     compute waves  iteration it: barrier; one row each (two for R = 16): header + entries out of the record,
//                    neighbour rows out of the LDS ring (ds_read_b128), fma in stored order (bit-identical to a
//                    sequential loop), store.  Their own vector-memory traffic (stores, row-local RK panels, which are
//                    requested one group ahead) is independent of the DMA waves' counters.
//   LDS: ring of D + 1 union buffers (CAP KiB each) + 2D + 1 records.  One workgroup barrier per group.
//
// Groups whose union exceeds CAP, whose entries exceed the record, or that hold a row longer than 64 entries are
// flagged by the plan; their rows are gathered directly from the CSR arrays by the compute waves (rare by
// construction: the plan is only attached when it covers the operator, csr_plan.hip: csr_create).
//
// Measured (spmm_lab, bit-exact): 8 consecutive lattice rows per group 0.467 ms = 4.55 TB/s algorithmic = 0.57 of
// the 8 TB/s peak; 4 x 4 lattice patches (16 rows, union 36) 0.407 ms = 5.22 TB/s = 0.65 - faster than the runtime's
// device-to-device copy of the same panel (0.43 ms).
#include <stdlib.h>

#include "kernels.h"
#include "rec_common.h"

#pragma clang fp contract(off)   // the RK algebra must round like the reference's separate mul / add ops

namespace ndcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRecWC = 8;        // compute waves
constexpr int kRecWD = 4;        // DMA waves
#ifndef NDCN_REC_D
#define NDCN_REC_D 2
#endif
constexpr int kRecD = NDCN_REC_D; // groups in flight ahead of the one being summed
constexpr int kRecMaxPrev = 5;

enum { REC_PLAIN = 0, REC_COMBINE = 1, REC_ERROR = 2, REC_RK4 = 3 };

struct RecArgs {
    const int *rec;
    int n_groups;
    const float *X, *Xh;
    int n_own;
    float *Y;
    float alpha;
    int relu;
    const int *rowptr, *colidx;      // direct gather of groups the plan could not stage
    const float *val;
    int dbg;                         // NDCN_REC_DBG (experiments): 1 request the panels in their own step, 2 wait for everything
};
struct RecEpi {
    const float *y0;
    const float *kprev[kRecMaxPrev];
    float *y_next;
    double *partials;                // ERROR: [gridDim.x * kRecWC][2]
    float c[kRecMaxPrev + 1];        // c[0 .. n_prev-1] for kprev, c[n_prev] for the new K ; RK4: c[0] = dt
    int n_prev;
    float rtol, atol;
    const float *y1;                 // ERROR: the state of the error record, by row of this launch
    float *y_aux;                    // COMBINE, nullable: second linear combination (no y0), coefficients c2[] (by value only)
    float c2[kRecMaxPrev + 1];
    const float *c_dev;              // nullable: the coefficients live in device memory instead of c[] (hipGraph replay: one
                                     // captured launch serves every step size; filled by scale_coef_kernel as fl(dt * c))
};

// row-local RK panels of one output row (requested one group ahead); MAXP = most earlier stages the mode can have
template <int MAXP> struct RecPanels { f32x4 km[MAXP]; f32x4 y0v, y1v; };

template <int R, int CAP, int RECW, bool HALO, int MODE>
__global__ __launch_bounds__(64 * (kRecWC + kRecWD)) void spmm_rec_kernel(RecArgs a, RecEpi e) {
    constexpr int D = kRecD, NBUF = D + 1, NREC = 2 * D + 1, CAPD = CAP / kRecWD, RPW = R / kRecWC, E0 = CAP + 2 * R;
    constexpr int MAXP = MODE == REC_RK4 ? 3 : kRecMaxPrev;
    constexpr bool WIDE = MODE == REC_PLAIN || RPW == 1;      // two rows per wave + their RK panels: keep the register budget
    typedef RecPanels<MAXP> Panels;
    static_assert(CAP % kRecWD == 0 && R % kRecWC == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_off = lane * 16;
    const unsigned lds_x = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    const unsigned lds_r = lds_x + NBUF * CAP * 1024;
    const int *rbuf = reinterpret_cast<const int *>(lds) + NBUF * CAP * 256;

    // groups of this workgroup: XCD x owns a contiguous chunk (its L2 sees neighbouring groups), its workgroups take
    // the chunk's groups round-robin
    const int xcd = blockIdx.x % kXcds, wg = blockIdx.x / kXcds, wpx = gridDim.x / kXcds;
    const int chunk = (a.n_groups + kXcds - 1) / kXcds;
    const int g_lo = xcd * chunk, g_hi = min(a.n_groups, g_lo + chunk);
    const int g0 = g_lo + wg;
    const int my = g0 < g_hi ? (g_hi - g0 + wpx - 1) / wpx : 0;

    if (wave < kRecWD) {
        // ------------------------------------------------------------------ DMA waves
        if (my == 0) return;
        const int d = wave;
        auto dma_rec = [&](int it) {
            if (d < RECW && it < my)
                dma_row(reinterpret_cast<const float *>(a.rec + ((size_t)(g0 + it * wpx) * RECW + d) * 256),
                        lds_r + (unsigned)(((it % NREC) * RECW + d) * 1024), lane_off);
        };
        auto dma_x = [&](int it) {
            const int *r = rbuf + (it % NREC) * RECW * 256 + d * CAPD;
            int cc[CAPD];
#pragma unroll
            for (int k = 0; k < CAPD; ++k) cc[k] = __builtin_amdgcn_readfirstlane(r[k]);
#pragma unroll
            for (int k = 0; k < CAPD; ++k) {
                const float *base = a.X;
                int c = cc[k];
                if (HALO && c >= a.n_own) { base = a.Xh; c -= a.n_own; }
                dma_row(base + (size_t)c * 256, lds_x + (unsigned)(((it % NBUF) * CAP + d * CAPD + k) * 1024), lane_off);
            }
        };
        for (int it = 0; it < 2 * D; ++it) dma_rec(it);
        rec_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                               // [P] the first records are readable
        for (int it = 0; it < D; ++it)
            if (it < my) dma_x(it);
        for (int it = 0; it < my; ++it) {
            // Group `it`'s rows (issued in iteration it - D, right after the record of group it + D) must have landed.
            // Vector memory completes in order; younger operations of this wave, in steady state: the D - 1
            // iterations of DMA issued since.
            if (it >= D && it + 2 * D <= my) {
                if (d < RECW) rec_wait_vmcnt<(D - 1) * (CAPD + 1)>();
                else rec_wait_vmcnt<(D - 1) * CAPD>();
            } else {
                rec_wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();                           // [A]
            dma_rec(it + 2 * D);
            if (it + D < my) dma_x(it + D);
        }
        rec_wait_vmcnt<0>();
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int cw = wave - kRecWD;
    double err_sum = 0.0, err_bad = 0.0;
    if (my == 0) {
        if (MODE == REC_ERROR && lane == 0) {                        // the finish kernel sums EVERY slot
            e.partials[2 * (blockIdx.x * kRecWC + cw)] = 0.0;
            e.partials[2 * (blockIdx.x * kRecWC + cw) + 1] = 0.0;
        }
        return;
    }
    const f32x4 *X = reinterpret_cast<const f32x4 *>(a.X);
    const f32x4 *Xh = reinterpret_cast<const f32x4 *>(a.Xh);
    f32x4 *Y = reinterpret_cast<f32x4 *>(a.Y);
    const int np = MODE == REC_PLAIN ? 0 : e.n_prev;
    auto coef = [&](int m) { return (MODE != REC_PLAIN && e.c_dev) ? e.c_dev[m] : e.c[m]; };    // wave-uniform (scalar loads)

    // Row-local RK panels are requested ONE GROUP AHEAD (the next group's row ids sit in its record, which landed D
    // iterations ago) from inline asm, so that hipcc's wait insertion - which cannot count requests issued under
    // run-time conditions and would fall back to vmcnt(0), i.e. wait for the requests of the NEXT group as well -
    // stays out of it.  s_cur / s_nxt count this wave's vector-memory operations issued after the requests of the
    // current / next group's panels: exactly that many may still be outstanding when the panels are consumed (vector
    // memory completes in order).
    int s_cur = 0, s_nxt = 0;
    auto hdr_row = [&](int it, int q) {
        return __builtin_amdgcn_readfirstlane(rbuf[(it % NREC) * RECW * 256 + CAP + 2 * (cw + kRecWC * q)]);
    };
    auto ldp = [&](const float *base /*uniform*/, int voff) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(voff), "s"(base) : "memory");
        return v;
    };
    auto request = [&](int row, Panels &p) {                      // panels < 4 GiB (launcher check)
        const int voff = (row << 10) + lane_off;
#pragma unroll
        for (int m = 0; m < MAXP; ++m)
            if (m < np) p.km[m] = ldp(e.kprev[m], voff);
        p.y0v = ldp(e.y0, voff);
        if (MODE == REC_ERROR) p.y1v = ldp(e.y1, voff);              // the input of this evaluation is y1 (own rows)
        s_cur += np + 1 + (MODE == REC_ERROR ? 1 : 0);
    };
    auto arrived = [&](Panels &p) {
#pragma unroll
        for (int m = 0; m < MAXP; ++m) asm volatile("" : "+v"(p.km[m]));
        asm volatile("" : "+v"(p.y0v));
        asm volatile("" : "+v"(p.y1v));
    };
    auto epilogue = [&](int row, f32x4 kn, const Panels &p) {
        const size_t o = (size_t)row * 64 + lane;
        f32x4 *yn = reinterpret_cast<f32x4 *>(e.y_next);
        if (MODE == REC_RK4) {
            // rk4_alt_step_func (rk_common.py:72-78), same operator order as fixed_stage_kernel ops 2-5
            const float dt = coef(0);
            f32x4 s;
            if (np == 0) s = (kn * dt) / 3.f;
            else if (np == 1) s = (p.km[0] / -3.f + kn) * dt;
            else if (np == 2) s = ((p.km[0] - p.km[1]) + kn) * dt;
            else s = (((p.km[0] + p.km[1] * 3.f) + p.km[MAXP > 2 ? 2 : 0] * 3.f) + kn) * (dt / 8.f);
            __builtin_nontemporal_store(p.y0v + s, yn + o);
            ++s_cur; ++s_nxt;
            return;
        }
        // sum of the stages left to right, the new one last (misc.py:22-25), each product rounded on its own
        f32x4 s = kn * coef(np);
        if (np > 0) {
            f32x4 u = p.km[0] * coef(0);
#pragma unroll
            for (int m = 1; m < MAXP; ++m)
                if (m < np) u = u + p.km[m] * coef(m);
            s = u + s;
        }
        if (MODE == REC_COMBINE) {
            __builtin_nontemporal_store(p.y0v + s, yn + o);
            ++s_cur; ++s_nxt;
            if (e.y_aux) {
                f32x4 w2 = kn * e.c2[np];
                if (np > 0) {
                    f32x4 u2 = p.km[0] * e.c2[0];
#pragma unroll
                    for (int m = 1; m < MAXP; ++m)
                        if (m < np) u2 = u2 + p.km[m] * e.c2[m];
                    w2 = u2 + w2;
                }
                __builtin_nontemporal_store(w2, reinterpret_cast<f32x4 *>(e.y_aux) + o);
                ++s_cur; ++s_nxt;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tol = e.atol + e.rtol * max_nan(fabsf(p.y0v[q]), fabsf(p.y1v[q]));
                const float z = s[q] / tol;
                err_sum += (double)(z * z);
                err_bad += (double)(int)(!(fabsf(p.y1v[q]) <= 3.402823466e38f));
            }
        }
    };

    // one group: `cur` holds the panels of this group's rows (requested during the previous group), `nxt` receives
    // those of the next group
    auto step = [&](int it, Panels (&cur)[RPW], Panels (&nxt)[RPW]) {
        __builtin_amdgcn_s_barrier();                               // [A] group `it` is staged
        if (MODE != REC_PLAIN && it + 1 < my) {
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int row = hdr_row(it + 1, q);
                if (row >= 0) request(row, nxt[q]);
            }
            s_nxt = 0;
            if (a.dbg & 1) rec_wait_vmcnt<0>();
        }
        const int *r = rbuf + (it % NREC) * RECW * 256;
        const f32x4 *xb = reinterpret_cast<const f32x4 *>(lds) + (it % NBUF) * CAP * 64;
        f32x4 kn[RPW];
        int rows[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int i = cw + kRecWC * q;
            const int row = __builtin_amdgcn_readfirstlane(r[CAP + 2 * i]);
            const int meta = __builtin_amdgcn_readfirstlane(r[CAP + 2 * i + 1]);
            rows[q] = row;
            kn[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row < 0) continue;
            const int cnt = meta & 0xffff, ofs = meta >> 16;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cnt != 0xffff) {
                int es = 0;
                float ev = 0.f;
                if (lane < cnt) { es = r[E0 + 2 * (ofs + lane)]; ev = __builtin_bit_cast(float, r[E0 + 2 * (ofs + lane) + 1]); }
                rec_row<WIDE>(es, ev, cnt, [&](int slot) { return xb[slot * 64 + lane]; }, acc);
            } else {
                // group not staged: gather this row from the CSR arrays, 64 entries at a time.  Loads from asm with a
                // full wait each round (rare path; keeps every vector load of these waves out of hipcc's bookkeeping)
                const int j0 = a.rowptr[row], j1 = a.rowptr[row + 1];
                for (int jb = j0; jb < j1; jb += 64) {
                    const int n = min(64, j1 - jb);
                    int es = 0;
                    float ev = 0.f;
                    if (lane < n) {
                        const int eo = (jb + lane) * 4;
                        asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(es), "=&v"(ev) : "v"(eo), "s"(a.colidx), "s"(a.val) : "memory");
                    }
                    rec_row<false>(es, ev, n, [&](int c) {
                        const float *p = a.X;
                        if (HALO && c >= a.n_own) { p = a.Xh; c -= a.n_own; }
                        f32x4 v;
                        asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(lane_off), "s"(p + (size_t)c * 256) : "memory");
                        return v;
                    }, acc);
                }
            }
            f32x4 o = acc * a.alpha;
            if (a.relu) { o.x = relu_nan(o.x); o.y = relu_nan(o.y); o.z = relu_nan(o.z); o.w = relu_nan(o.w); }
            __builtin_nontemporal_store(o, &Y[(size_t)row * 64 + lane]);
            ++s_cur; ++s_nxt;
            kn[q] = o;
        }
        if (MODE != REC_PLAIN) {
            // cur's requests are complete once at most s_cur operations are outstanding: ONE wait per group at a fixed
            // program point (tools/audit_async_regs.py follows every path from a request to it)
            rec_wait_vmcnt_rt((a.dbg & 2) ? 0 : s_cur);
#pragma unroll
            for (int q = 0; q < RPW; ++q) arrived(cur[q]);
#pragma unroll
            for (int q = 0; q < RPW; ++q)
                if (rows[q] >= 0) epilogue(rows[q], kn[q], cur[q]);
        }
        s_cur = s_nxt;
    };
    Panels pa[RPW], pb[RPW];
    __builtin_amdgcn_s_barrier();                                   // [P]
    if (MODE != REC_PLAIN) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int row = hdr_row(0, q);
            if (row >= 0) request(row, pa[q]);
        }
        s_cur = 0;
    }
    for (int it = 0; it < my; it += 2) {
        step(it, pa, pb);
        if (it + 1 < my) step(it + 1, pb, pa);
    }
    if (MODE != REC_PLAIN) {
        // nothing is in flight here (the last group requests nothing); said explicitly so that EVERY path from a request
        // reaches a wait before the registers can be touched again (tools/audit_async_regs.py)
        rec_wait_vmcnt<0>();
#pragma unroll
        for (int q = 0; q < RPW; ++q) { arrived(pa[q]); arrived(pb[q]); }
    }
    if (MODE == REC_ERROR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            err_sum += __shfl_down(err_sum, off, 64);
            err_bad += __shfl_down(err_bad, off, 64);
        }
        if (lane == 0) {
            e.partials[2 * (blockIdx.x * kRecWC + cw)] = err_sum;
            e.partials[2 * (blockIdx.x * kRecWC + cw) + 1] = err_bad;
        }
    }
}

static int env_int_rec(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

int spmm_rec_supported(const ndcn_csr *A, int H) {
    static const int enabled = env_int_rec("NDCN_SPMM_REC", 1);
    if (!enabled || H != 256 || !A || !A->rec || A->rec_groups <= 0) return 0;
    return (A->rec_rows == 8 && A->rec_cap == 32 && A->rec_kib == 1) || (A->rec_rows == 16 && A->rec_cap == 40 && A->rec_kib == 2) ||
           (A->rec_rows == 8 && A->rec_cap == 48 && A->rec_kib == 2);      // ring + shortcuts: small-world graphs
}

int spmm_rec_variant(int mode, int n_prev) {
    if (mode == REC_PLAIN) return 1;
    if (mode == REC_COMBINE) return n_prev >= 0 && n_prev <= kRecMaxPrev;
    if (mode == REC_RK4) return n_prev >= 0 && n_prev <= 3;
    return mode == REC_ERROR && n_prev >= 0 && n_prev <= kRecMaxPrev;
}

int64_t spmm_rec_partials_bytes() { return (int64_t)kCus * kRecWC * 2 * sizeof(double); }

template <int R, int CAP, int RECW>
static int launch_rec(const RecArgs &a, const RecEpi &e, int mode, bool halo, hipStream_t st, dim3 &grid_out) {
    constexpr size_t lds = (size_t)((kRecD + 1) * CAP + (2 * kRecD + 1) * RECW) * 1024;
    int per_xcd = kCus / kXcds;
    const int need = (a.n_groups + kXcds - 1) / kXcds;
    if (per_xcd > need) per_xcd = need;
    const dim3 grid(per_xcd * kXcds), block(64 * (kRecWC + kRecWD));
    grid_out = grid;
#define NDCN_REC(HALO_, MODE_)                                                                                        \
    do {                                                                                                              \
        auto kern = spmm_rec_kernel<R, CAP, RECW, HALO_, MODE_>;                                                      \
        static std::atomic<unsigned long long> attr_seen{0};                                                                                 \
        if (once_per_device(attr_seen)) {                                                                                              \
            NDCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a, e);                                                         \
    } while (0)
#define NDCN_REC_MODE(HALO_)                                     \
    do {                                                         \
        if (mode == REC_PLAIN) NDCN_REC(HALO_, REC_PLAIN);       \
        else if (mode == REC_COMBINE) NDCN_REC(HALO_, REC_COMBINE); \
        else if (mode == REC_ERROR) NDCN_REC(HALO_, REC_ERROR);  \
        else NDCN_REC(HALO_, REC_RK4);                           \
    } while (0)
    if (halo) NDCN_REC_MODE(true);
    else NDCN_REC_MODE(false);
#undef NDCN_REC_MODE
#undef NDCN_REC
    NDCN_LAUNCH_CHECK();
    return NDCN_OK;
}

// mode 0: Y = alpha (A X) [relu];  modes 1-3: K = relu(A X) plus the RK algebra (see rhs_fused2_f32)
int spmm_rec_f32(const ndcn_csr *A, const float *X, const float *Xh, int64_t n_own, float *Y, float alpha, uint32_t flags,
                 int mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev, float *y_next,
                 float rtol, float atol, double *d_out, void *d_ws, hipStream_t st, const float *c_dev, const RkOpt *opt) {
    if (A->n_rows == 0) return NDCN_OK;
    if (!spmm_rec_variant(mode, n_prev)) { set_error("spmm_rec: no kernel for mode %d with %d previous stages", mode, n_prev); return NDCN_EINVAL; }
    RecArgs a;
    a.rec = A->rec; a.n_groups = A->rec_groups; a.X = X; a.Xh = Xh; a.n_own = (int)n_own; a.Y = Y; a.alpha = alpha;
    a.relu = (flags & NDCN_F_RELU) ? 1 : 0;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
    a.dbg = env_int_rec("NDCN_REC_DBG", 0);
    RecEpi e = {};
    e.y0 = y0; e.y_next = y_next; e.n_prev = n_prev; e.rtol = rtol; e.atol = atol; e.partials = static_cast<double *>(d_ws);
    e.c_dev = c_dev;
    e.y1 = (opt && opt->y1) ? opt->y1 : X;
    e.y_aux = (mode == REC_COMBINE && !c_dev && opt && opt->y_aux && opt->c_aux) ? opt->y_aux : nullptr;
    for (int m = 0; m <= kRecMaxPrev; ++m) e.c2[m] = (e.y_aux && m <= n_prev) ? opt->c_aux[m] : 0.f;
    for (int m = 0; m < kRecMaxPrev; ++m) e.kprev[m] = (m < n_prev && h_kprev) ? h_kprev[m] : nullptr;
    for (int m = 0; m <= kRecMaxPrev; ++m) e.c[m] = (mode != REC_PLAIN && mode != REC_RK4 && m <= n_prev) ? h_c[m] : 0.f;
    if (mode == REC_RK4) e.c[0] = h_c[0];
    const double P = 4.0 * 256 * (double)A->n_rows;
    double bytes = 8.0 * A->nnz + 4.0 * (A->n_rows + 1) + 4.0 * 256 * (double)(A->n_rows + A->n_cols);
    if (mode != REC_PLAIN) bytes += P * (n_prev + 2 + (e.y_aux ? 1 : 0));
    ProfScope prof(mode == REC_PLAIN ? PROF_SPMM : PROF_RHS_FUSED, st, bytes, 2.0 * A->nnz * 256);
    dim3 grid;
    int rc;
    if (A->rec_rows == 8 && A->rec_cap == 48) rc = launch_rec<8, 48, 2>(a, e, mode, Xh != nullptr, st, grid);
    else if (A->rec_rows == 8) rc = launch_rec<8, 32, 1>(a, e, mode, Xh != nullptr, st, grid);
    else rc = launch_rec<16, 40, 2>(a, e, mode, Xh != nullptr, st, grid);
    if (rc) return rc;
    if (mode == REC_ERROR) return partials_finish(e.partials, (int)grid.x * kRecWC, d_out, st, (opt && opt->accum) ? 1 : 0);
    return NDCN_OK;
}

}  // namespace ndcn
