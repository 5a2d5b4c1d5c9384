/*
 * ndcn_hip.h - C ABI of libndcn_hip.so: the MI355X (gfx950) implementation of the NDCN ODEFunc hot path.
 *
 * The reference (calvin-zcx/ndcn) is pure Python and has no FFI of its own; its hot path is a set of
 * PyTorch op call sites.  Each entry point below replaces one of those call sites (or one
 * solver-side group of them) and is what a binding of the reference for this path would call.  The
 * reference-side ctypes stub is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ (host) ;
 *   - panels are dense row-major fp32, n_rows x H, H floats per node; 16-byte aligned base;
 *   - CSR operators are {rowptr int32[n_rows+1], colidx int32[nnz], val fp32[nnz]}, columns of a row
 *     in ascending order (summation order = stored order);
 *   - `stream` is a hipStream_t (passed as void*; NULL = the null stream); calls only ENQUEUE work, they
 *     never synchronise, allocate or free device memory, except where stated (ndcn_solver_*);
 *   - return value: 0 on success, a negative NDCN_E* code otherwise; ndcn_last_error() gives the text.
 *     Nothing throws across the boundary.
 *   - the library never falls back to a host computation.
 *
 * Reference paths are relative to the reference repository root.
 */
#ifndef NDCN_HIP_H
#define NDCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDCN_ABI_VERSION 18
#define NDCN_API __attribute__((visibility("default")))

#define NDCN_OK          0
#define NDCN_EINVAL     -1   /* bad argument (null pointer, misaligned panel, unsupported H, ...) */
#define NDCN_EHIP       -2   /* a HIP runtime call failed; see ndcn_last_error() */
#define NDCN_ENONFINITE -3   /* solver: non-finite state  (dopri5.py:101-102 assertion)          */
#define NDCN_EUNDERFLOW -4   /* solver: t0 + dt <= t0     (dopri5.py:100 assertion)              */
#define NDCN_EMAXSTEPS  -5   /* solver: max_num_steps exceeded (dopri5.py:89 assertion)          */
#define NDCN_ESTATE     -6   /* solver handle used out of order                                   */

/* rhs / spmm flags */
#define NDCN_F_RELU        1u   /* apply relu to the result            (neural_dynamics.py:36)   */
#define NDCN_F_NO_GRAPH    2u   /* skip A*x                            (neural_dynamics.py:27)   */
#define NDCN_F_NO_CONTROL  4u   /* skip the Linear                     (neural_dynamics.py:32)   */
#define NDCN_F_PACKED      8u   /* ndcn_rhs_f32 / ndcn_rhs_rk_f32, H = 256: `work` still holds the packed image of W that an
                                 * earlier call on this stream (flag clear, same W contents) left there - skip the re-pack   */
#define NDCN_F_ACCUM      16u   /* ndcn_rhs_rk_f32, NDCN_RK_ERROR: ADD this launch's {sum, bad} to d_out instead of overwriting
                                 * it - an evaluation split into several launches on one stream (row blocks of a shard)      */

/* integrator methods (torchdiffeq/_impl/odeint.py:8-17, the in-scope subset) */
#define NDCN_M_EULER    0
#define NDCN_M_MIDPOINT 1
#define NDCN_M_RK4      2   /* the 3/8 rule, rk_common.py:72-78 */
#define NDCN_M_DOPRI5   3

typedef struct ndcn_csr {
    int64_t n_rows;
    int64_t n_cols;           /* columns >= n_own address the halo panel (see ndcn_spmm_f32)     */
    int64_t nnz;
    const int32_t *rowptr;    /* [n_rows + 1] */
    const int32_t *colidx;    /* [nnz]        */
    const float   *val;       /* [nnz]        */
    const int32_t *row_order; /* [n_rows] or NULL: a permutation of the rows giving the order in which the
                                 kernels WALK them (a cache-locality hint, e.g. lattice tiles); results are
                                 identical for any permutation                                              */
    const int32_t *tile_order; /* [ceil(n_rows / 64)] or NULL: the order in which the fused RHS kernel walks its 64-row
                                 tiles (a permutation; locality hint like row_order - results are identical)      */
    /* Optional "group record" plan (rec = NULL when absent), built once per operator by ndcn_csr_create
     * (ndcn_amd/csrc/csr_plan.hip) for H = 256 panels.  The rows are cut into groups of rec_rows rows that are
     * consecutive in the operator's walk order (row_order, or 0..n-1); group g owns the fixed-size record
     * rec[g * rec_kib * 256 ...) of 32-bit words:
     *   [0, rec_cap)                    the DISTINCT columns the group's rows reference, ascending, padded by repetition
     *   [rec_cap, rec_cap + 2 rec_rows) per row {row id or -1, cnt | ofs << 16}; cnt = 0xffff flags a group the record
     *                                   cannot hold (the kernel gathers its rows from rowptr / colidx / val directly)
     *   [rec_cap + 2 rec_rows, ...)     the rows' entries as (index into the group's column list, fp32 value bits), row
     *                                   after row in stored (column-ascending) order
     * The SpMM kernel moves every record and every distinct neighbour row into LDS by DMA at addresses it can form
     * without first loading streamed index data (ndcn_amd/csrc/spmm_rec.hip).  Supported shapes:
     * {rec_rows, rec_cap, rec_kib} = {8, 32, 1}, {16, 40, 2} (lattice patches) and {8, 48, 2} (8 consecutive rows of a
     * ring-plus-shortcuts graph: 12 ring columns + ~2 shortcut endpoints per row).                            */
    int32_t        rec_rows, rec_cap, rec_kib, rec_groups;
    const int32_t *rec;       /* [rec_groups][rec_kib * 256] */
    /* Optional long-row plan (hub_n = 0 when absent), built once per operator by ndcn_csr_create
     * (ndcn_amd/csrc/csr_plan.hip) for graphs with a skewed degree distribution.  Rows with more than the
     * plan's threshold of entries ("hubs": a 3900-entry row of a 10^6-node Barabasi-Albert graph is otherwise
     * one wave's sequential work inside the fused RHS kernel) are evaluated ahead of that kernel:
     *   hub_seg_* : CSR over SEGMENTS of the hub rows (<= 256 entries each; columns as in colidx),
     *               S_seg = hub_seg x X                         -> hub_Sseg [hub_nseg][H]
     *   hub_cmb_* : CSR [hub_n x hub_nseg] of ones, row h = the segments of hub h in order,
     *               S_hub = hub_cmb x S_seg                     -> hub_S [hub_n][H]
     *   lt_*      : the operator with every hub row replaced by ONE entry (column n_cols + h, value 1): the
     *               fused kernel reads S_hub as its second ("halo") panel.
     * hub_S / hub_Sseg are scratch owned by the operator (one launch stream at a time).  Used by ndcn_rhs_f32 /
     * ndcn_rhs_rk_f32 / the solver when H == hub_H and either no halo panel is passed or hub_S lies directly behind
     * the halo panel in ONE allocation (hub_S == X_halo + n_halo * H: the kernel's second panel is then
     * [halo | hubs]; node-range shards of power-law graphs, ndcn_amd/sharding.py).                              */
    int32_t        hub_n, hub_nseg, hub_H;
    int64_t        hub_nnz, lt_nnz;
    const int32_t *hub_seg_rowptr;  /* [hub_nseg + 1] */
    const int32_t *hub_colidx;      /* [hub_nnz] */
    const float   *hub_val;         /* [hub_nnz] */
    const int32_t *hub_cmb_rowptr;  /* [hub_n + 1] */
    const int32_t *hub_cmb_colidx;  /* [hub_nseg] = 0 .. hub_nseg-1 */
    const float   *hub_cmb_val;     /* [hub_nseg] ones */
    const int32_t *lt_rowptr;       /* [n_rows + 1] */
    const int32_t *lt_colidx;       /* [lt_nnz] */
    const float   *lt_val;          /* [lt_nnz] */
    float         *hub_Sseg;        /* [hub_nseg][hub_H] */
    float         *hub_S;           /* [hub_n][hub_H] */
    /* Facts about the arrays that some entry points need and would otherwise read back from the device at every call
     * (ndcn_solve_small_*: the ELL width; its reverse sweep: whether A^T = A).  0 = not known: the library finds out, with one
     * small device-to-host copy per call.  ndcn_csr_create fills max_row_len; the Python binding fills both once per operator. */
    int32_t        max_row_len;     /* the longest row, or 0 */
    int32_t        symmetric;       /* 1: the stored arrays equal those of the transpose; 2: they do not; 0: unknown */
    /* Optional column-sweep plan (sweep_ent = NULL when absent), built by ndcn_csr_create for H = 256 operators WITHOUT
     * locality whose rows are long relative to their count (a 10^5-node G(n,p) graph of mean degree 40: heat_dynamics.py:89 at
     * BASELINE config 2).  A row gather fetches nnz rows of X through the fabric (L2 hit rate 6 %); the sweep keeps the partial
     * sums of ALL rows in registers and lets every XCD walk the columns of X in ascending order, so that the rows of X an XCD
     * needs at one time form a window of its 4 MiB L2: 8 * n_cols rows cross the fabric per pass instead of nnz
     * (ndcn_amd/csrc/spmm_sweep.hip).  Layout: the rows are cut into sweep_passes passes of <= 100 352 rows; a pass gives each of
     * its 8 x 256 waves a "slab" of sweep_rpw consecutive rows (8 XCD chunks of ceil(rows / 8) rows, 256 slabs per chunk);
     *   sweep_slab [passes * 2048][2]  {first entry (a multiple of 8), entries} of the slab in sweep_ent
     *   sweep_ent  pairs of 32-bit words {row within the slab << 24 | column, fp32 value bits}: the slab's entries merged over
     *              its rows and sorted by column - per row still in ascending column order, i.e. the fma chain of a sequential
     *              CSR loop: results are bit-identical to it - padded to groups of 8 with {49 << 24, 0}
     *   sweep_prog [passes][8][256] + [passes]   one word per wave: (launch tag << 16) | column block it fetches from, then one
     *              launch counter per pass (the tag's source: kept on the device so that a hipGraph replay advances it) - waves of an XCD
     *              keep within sweep_window blocks of 2^sweep_logb columns of each other (a locality hint with a bounded wait,
     *              never needed for correctness)
     *   sweep_S    [n_rows][256] scratch for S = A X in front of the Linear (ndcn_rhs_f32 / ndcn_rhs_rk_f32: the fused kernel
     *              then runs on the identity operator sweep_eye_* over S, whose 16-row group records stage S by LDS-DMA:
     *              rhs_fused3) - owned by the operator, like hub_S: one launch stream at a time.  Used when no halo panel is
     *              passed.                                                                                                 */
    int32_t         sweep_passes, sweep_rpw, sweep_logb, sweep_window;
    int64_t         sweep_rows_per_pass;
    const uint32_t *sweep_ent;
    const int32_t  *sweep_slab;
    uint32_t       *sweep_prog;
    float          *sweep_S;
    const int32_t  *sweep_eye_rowptr;  /* [n_rows + 1] = 0 .. n_rows */
    const int32_t  *sweep_eye_colidx;  /* [n_rows] = 0 .. n_rows - 1 */
    const float    *sweep_eye_val;     /* [n_rows] ones */
    const int32_t  *sweep_eye_rec;     /* the identity operator's group-record plan {16, 40, 2}: the dense stage runs rhs_fused3 */
    int32_t         sweep_eye_groups;
} ndcn_csr;

/* ------------------------------------------------------------------------------------------------
 * Operator handle.  The reference holds its operator `A` by reference on the module (neural_dynamics.py:9-18: `self.A = A`)
 * and the caller converts it once, before model construction (heat_dynamics.py:170-175: dense -> sparse COO, .to(device)).
 * The counterpart: hand the CSR arrays over ONCE and keep the handle next to the model.  ndcn_csr_create builds - on the
 * device, from the arrays alone - every plan the H = 256 kernels select on (struct ndcn_csr above: the group-record plan
 * with its lattice walk orders, the long-row plan), so a caller that binds nothing but this header reaches the same kernels
 * (rhs_fused3, spmm_rec) as the Python package, whose ndcn_amd/csr.py is a binding of these calls.
 *   rowptr / colidx / val   DEVICE arrays owned by the caller; they must outlive the handle (the view points into them)
 *   H                       panel width the operator will be applied to (plans exist for H = 256; other widths: a bare view)
 *   hints                   nullable; zero-initialise, then set what applies:
 *     lattice_row_base / lattice_n_own   this operator is a ROW BLOCK of a node-range shard: row r is node row_base + r of the
 *                           shard, columns < n_own are the shard's own nodes, columns >= n_own halo rows (lattice_n_own = 0:
 *                           a whole graph; stencil detection then requires a square operator)
 *     n_halo                rows of the halo panel that will be passed next to X: the long-row plan lays the hubs' scratch
 *                           rows out directly behind them ([halo | hub rows], ndcn_csr_halo_panel)
 *     row_order             [n_rows] walk order given by the caller (the view's row_order; also groups the records)
 *     group_order / n_group_order   row ids (-1 = empty slot) in the order the records group them, e.g. a lattice's patches
 *     rec_rows / rec_cap / rec_kib  != 0: build exactly this record shape and keep it whatever it covers (default: the
 *                           shapes are tried and one is kept only if it holds >= 90 % of the entries and stages <= 3/4 of
 *                           the rows a direct gather would fetch)
 *     hub_threshold         > 0: rows longer than this leave the fused kernel; 0: automatic (the lowest of 32 / 64 / 128 that
 *                           moves <= 5 % of the rows); < 0: no long-row plan
 *     flags                 NDCN_PLAN_*
 * Allocates the plans with hipMalloc (synchronises); every other call takes ndcn_csr_view(handle).  `stream` orders the
 * plan kernels; the call returns after they have finished.                                                              */
typedef struct ndcn_csr_handle ndcn_csr_handle;
typedef struct ndcn_csr_hints {
    int64_t        lattice_row_base, lattice_n_own;
    int64_t        n_halo;
    const int32_t *row_order;
    const int32_t *group_order;
    int64_t        n_group_order;
    int32_t        rec_rows, rec_cap, rec_kib;
    int32_t        hub_threshold;
    uint32_t       flags;
} ndcn_csr_hints;
#define NDCN_PLAN_NO_REC            1u   /* no group-record plan                                                     */
#define NDCN_PLAN_NO_STENCIL        2u   /* do not look for a lattice stencil (records group consecutive rows)        */
#define NDCN_PLAN_NO_TILE_ORDER     4u   /* no tile walk order for the fused kernel                                   */
#define NDCN_PLAN_NO_HUB            8u   /* no long-row plan                                                          */
#define NDCN_PLAN_EXTERNAL_SCRATCH 16u   /* the caller provides the long-row plan's scratch (ndcn_csr_set_hub_scratch) */
#define NDCN_PLAN_ORDER_ONLY       32u   /* lattice detection and walk orders only, no records                        */
#define NDCN_PLAN_NO_SWEEP         64u   /* no column-sweep plan                                                      */
#define NDCN_PLAN_FORCE_SWEEP     128u   /* build the column-sweep plan whatever the fetch arithmetic says (tests)     */
NDCN_API int ndcn_csr_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr, const int32_t *colidx,
                             const float *val, int H, const ndcn_csr_hints *hints, void *stream, ndcn_csr_handle **out);
NDCN_API int ndcn_csr_destroy(ndcn_csr_handle *h);
/* the struct every other entry point takes; valid until ndcn_csr_destroy */
NDCN_API const ndcn_csr *ndcn_csr_view(const ndcn_csr_handle *h);
/* h_out = {rec_rows, rec_cap, rec_kib, rec_groups, entries held by records, distinct columns staged by them, lattice stride
 * (0: none), slots of the group order, hub_n, hub_nseg, hub threshold, hub_nnz, lt_nnz, n_halo, has tile order, H}     */
NDCN_API int ndcn_csr_info(const ndcn_csr_handle *h, int64_t h_out[16]);
/* the group order the records were built on when the library found it (a detected lattice): [info[7]] int32, or NULL    */
NDCN_API const int32_t *ndcn_csr_group_order(const ndcn_csr_handle *h);
/* long-row plan: the [n_halo + hub_n][H] buffer whose head receives the halo rows and whose tail holds the hubs' rows - pass
 * it as X_halo (NULL without a long-row plan, or before ndcn_csr_set_hub_scratch under NDCN_PLAN_EXTERNAL_SCRATCH)        */
NDCN_API float *ndcn_csr_halo_panel(const ndcn_csr_handle *h);
/* NDCN_PLAN_EXTERNAL_SCRATCH: Sseg [hub_nseg][H] and halo_S [n_halo + hub_n][H], 16-byte aligned, owned by the caller     */
NDCN_API int ndcn_csr_set_hub_scratch(ndcn_csr_handle *h, float *Sseg, float *halo_S);
/* column-sweep plan: h_out = {passes, rows per wave, log2 of the column block, window in blocks, padded entries, rows per pass,
 * 1 if the scratch panel is set, 0} (all zero without the plan); under NDCN_PLAN_EXTERNAL_SCRATCH the caller provides the
 * [n_rows][256] scratch panel (16-byte aligned) before the first right-hand side                                            */
NDCN_API int ndcn_csr_sweep_info(const ndcn_csr_handle *h, int64_t h_out[8]);
NDCN_API int ndcn_csr_set_sweep_scratch(ndcn_csr_handle *h, float *S);

NDCN_API int         ndcn_abi_version(void);
NDCN_API const char *ndcn_last_error(void);            /* thread-local, valid until the next failing call */
/* {multiProcessorCount, 8 XCDs assumed, warpSize, clock kHz, totalGlobalMem>>20, l2CacheSize>>10}      */
NDCN_API int         ndcn_device_info(int64_t h_out[6]);

/* ------------------------------------------------------------------------------------------------
 * Y = alpha * (A X)   [then relu if NDCN_F_RELU]
 * replaces torch.sparse.mm(A, x)  neural_dynamics.py:29 (and torch.mm(A, x) :31 for a dense A held as
 * CSR); ode_gcn.py:53; models.py:18; heat_dynamics.py:201 (alpha = -k, H = 1).
 * X_halo (nullable): rows of remote nodes; a column c >= n_own reads X_halo[c - n_own] (multi-GPU
 * node-range sharding). With X_halo == NULL every column must be < n_own == A->n_cols.
 * Y must not alias X.
 */
NDCN_API int ndcn_spmm_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own,
                  float *Y, int H, float alpha, uint32_t flags, void *stream);

/* Y[n, H_out] = S[n, H_in] W^T + b  [then relu]; W is nn.Linear's [H_out, H_in] row-major, b nullable.
 * replaces self.wt(x) neural_dynamics.py:33 ; the encoder/decoder Linears :143-148 ; models.py:15.
 * Y must not alias S. */
NDCN_API int ndcn_linear_f32(const float *S, const float *W, const float *b, float *Y,
                    int64_t n, int H_in, int H_out, uint32_t flags, void *stream);

/* GraphConvolution.forward in the reference's own order (models.py:14-18; neural_dynamics.py:171-176): support = X W^T + b, then
 * Y = A support  [then relu] - i.e. ndcn_linear_f32 followed by ndcn_spmm_f32 in ONE call (two launches on `stream`).
 * X [n_cols(A), H_in], Y [n_rows(A), H_out]; work: ndcn_gcn_work_bytes() bytes of device scratch (the support panel).       */
NDCN_API int ndcn_gcn_f32(const ndcn_csr *A, const float *X, const float *W, const float *b, float *Y, float *work,
                          int H_in, int H_out, uint32_t flags, void *stream);
NDCN_API int64_t ndcn_gcn_work_bytes(int64_t n_cols, int H_out);

/* Backward of Y = act(S W^T + b) - the reference trains by plain autograd through every solver op
 * (heat_dynamics.py:333, dgnn.py:204; neural_dynamics.py:33,36,143-148).  gZ = g (.) [Y > 0] when the ReLU output Y is
 * given (nullable: no activation), then
 *   gS [n, H_in]      = gZ W               (nullable)
 *   gW [H_out, H_in]  = gZ^T S             (nullable; reduction over rows split into chunks, partials summed in a fixed
 *                                           order: deterministic, no atomics)
 *   gb [H_out]        = column sums of gZ  (nullable)
 * S: the forward input (needed for gW).  work: ndcn_linear_bwd_work_bytes() bytes of device scratch (gW / gb; with
 * H_in = H_out = 256 also gS: given scratch, the product runs on the fp16 matrix cores from two-piece splits of both
 * operands - fp32-grade, as the forward kernels - and the planes of W^T are packed there; without it, the fp32 MFMA).   */
NDCN_API int ndcn_linear_bwd_f32(const float *g, const float *Y, const float *S, const float *W, float *gS, float *gW,
                                 float *gb, void *work, int64_t n, int H_in, int H_out, uint32_t flags, void *stream);
/* flags: NDCN_F_PACKED (H_in = H_out = 256, gS) - `work` still holds the planes of W^T that an earlier call with the same W and
 * the same n packed there (its tail; the head is this call's scratch): the backward of a solve evaluates the right-hand side's
 * VJP dozens of times between two weight updates.                                                                          */
NDCN_API int64_t ndcn_linear_bwd_work_bytes(int64_t n, int H_in, int H_out);
/* out = w * x: the VJP of one term of the Runge-Kutta linear combinations (rk_common.py:51,75-78; interp.py:21-35). */
NDCN_API int ndcn_scale_f32(float *out, const float *x, float w, int64_t n_elem, void *stream);
/* out = g where y > 0, else 0: the VJP of relu given its OUTPUT y (neural_dynamics.py:36; the no_control RHS). */
NDCN_API int ndcn_relu_bwd_f32(float *out, const float *g, const float *y, int64_t n_elem, void *stream);
/* dst = src, the library's plain streaming pass (16 bytes per lane, non-temporal): what `x.clone()` / the solver's state
 * hand-overs cost, and the measured HBM ceiling bench.py quotes next to the 8 TB/s spec peak.                            */
NDCN_API int ndcn_copy_f32(float *dst, const float *src, int64_t n_elem, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Vector-Jacobian products of the dopri5 panel operations.  The reference's vendored torchdiffeq differentiates through
 * its step-size controller - dt, dt * beta, the error ratio, the initial-step norms and the interpolation abscissa are
 * tensors with autograd history (rk_common.py:41-61, misc.py:84-170, interp.py:38-65) - and the drivers train by plain
 * backprop through the solver.  Each call is ONE pass: every panel gradient of the operation (h_gk / g* entries may be
 * NULL) plus, in d_dots[0..7] (fp64, fixed-order sums), the inner products that become the gradients of the scalar
 * inputs.  d_ws: ndcn_rk_bwd_ws_bytes() bytes of device scratch.
 *   combine   (misc.py:22-25)     y = y0 + sum_j c_j k_j :  gk_j = c_j g ;  d_dots[j] = <g, k_j> (= g_cj) ;  g_y0 = g
 *   error     (misc.py:146-157)   r = mean((sum_j c_j k_j / tol)^2), tol = atol + rtol max(|y0|, |y1|), upstream g_r,
 *                                 inv_n = 1 / numel :  gk_j, gy0, gy1 ;  d_dots[j] = d r / d c_j (multiply by g_r)
 *   rms       (misc.py:71-76)     o = ||(a - b) / (atol + |y| rtol)||_2 / sqrt(N), coef = g_o / (||.|| sqrt(N)) :  ga, gb, gy
 *   interp    (dopri5.py:39-45, interp.py:21-65)  o = dense output at abscissa x for step size dt :  gy0, gy1, gk_0..6 ;
 *                                 d_dots[0] = <g, d o / d x>, d_dots[1] = <g, d o / d dt>                              */
NDCN_API int64_t ndcn_rk_bwd_ws_bytes(void);
/* Accumulating form (h_acc / acc_y0 / acc_y1, all nullable, entries nullable): a panel that feeds SEVERAL later operations - every
 * stage derivative of a Runge-Kutta step does - receives one gradient per consumer, which autograd would add up with a pass of its
 * own each (`add` kernels were 24 % of a dopri5 training step).  Given the gradient such a panel has ALREADY received (acc), the
 * kernels write  gk_j = acc_j + (this operation's contribution)  in the pass they make anyway; outputs may alias their acc.
 * combine: gy0 = acc_y0 + g is written only when acc_y0 is given (without one, g_y0 is g itself).                                */
NDCN_API int ndcn_rk_combine_bwd_f32(const float *g, const float *const *h_k, const float *h_c, int n_k, float *const *h_gk,
                                     const float *const *h_acc, float *gy0, const float *acc_y0, double *d_dots, void *d_ws,
                                     int64_t n_elem, void *stream);
/* d_dots[0] = <g, a - b> (b nullable: <g, a>), fp64 partial sums in a fixed order.  For a stage sum u = y0 + dt sum_j beta_j k_j the
 * gradient of dt is <g_u, u - y0> / dt - three panels read instead of one per term (rk_common.py:41-51 differentiated).            */
NDCN_API int ndcn_rk_dot_diff_f32(const float *g, const float *a, const float *b, double *d_dots, void *d_ws, int64_t n_elem,
                                  void *stream);
NDCN_API int ndcn_rk_error_bwd_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k,
                                   float rtol, float atol, float g_r, double inv_n, float *gy0, float *gy1,
                                   float *const *h_gk, const float *acc_y0, const float *acc_y1, const float *const *h_acc,
                                   double *d_dots, void *d_ws, int64_t n_elem, void *stream);
NDCN_API int ndcn_rk_rms_bwd_f32(const float *a, const float *b, const float *y, float rtol, float atol, float coef, float *ga,
                                 float *gb, float *gy, int64_t n_elem, void *stream);
NDCN_API int ndcn_dopri5_interp_bwd_f32(const float *g, const float *y0, const float *y1, const float *const *h_k /*7*/,
                                        float dt, float x, float *gy0, float *gy1, float *const *h_gk /*7*/, const float *acc_y0,
                                        const float *acc_y1, const float *const *h_acc /*7*/, double *d_dots, void *d_ws,
                                        int64_t n_elem, void *stream);

/* n_t <= 7 ticks of ONE accepted step in one pass: the forward (fit + evaluate, as ndcn_dopri5_interp_direct_f32 per tick - the same
 * bits) and its VJP (as ndcn_dopri5_interp_bwd_f32 summed over the ticks; d_dots[t] = <g_t, d o / d x_t>, d_dots[7] = sum_t <g_t,
 * d o / d dt>).  The drivers sample 16-120 ticks over a handful of steps (heat_dynamics.py:35,123; dgnn.py:173-182): the step's
 * panels are read once per launch instead of once per tick.  h_out / h_g: HOST arrays of n_t device panels; h_xpow: n_t x 5.       */
NDCN_API int ndcn_dopri5_interp_direct_multi_f32(const float *y0, const float *y1, const float *const *h_k /*7*/,
                                                 const float *h_cmid /*7*/, float dt, const float *h_xpow, float *const *h_out,
                                                 int n_t, int64_t n_elem, void *stream);
NDCN_API int ndcn_dopri5_interp_bwd_multi_f32(const float *const *h_g, int n_t, const float *y0, const float *y1,
                                              const float *const *h_k /*7*/, float dt, const float *h_x, float *gy0, float *gy1,
                                              float *const *h_gk /*7*/, const float *acc_y0, const float *acc_y1,
                                              const float *const *h_acc /*7*/, double *d_dots, void *d_ws, int64_t n_elem,
                                              void *stream);

/* The whole ODEFunc.forward in one call: Y = relu(W (A X) + b) honouring NO_GRAPH / NO_CONTROL
 * (neural_dynamics.py:20-39, dropout p = 0).  `work`: device scratch of ndcn_rhs_work_bytes() bytes, 16-byte
 * aligned (H = 256: the fused SpMM->LDS->MFMA kernel keeps its packed weights there; other widths: the
 * S = A X panel between the SpMM and the Linear kernel; 0 bytes when the Linear or the SpMM is skipped). */
NDCN_API int ndcn_rhs_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own,
                 const float *W, const float *b, float *Y, float *work, int H, uint32_t flags, void *stream);
NDCN_API int64_t ndcn_rhs_work_bytes(int64_t n_rows, int H, uint32_t flags);

/* The right-hand side of the ADJOINT system for ODEFunc (odeint_adjoint: torchdiffeq/_impl/adjoint.py:34-59, where the reference
 * evaluates func under enable_grad and calls torch.autograd.grad with cotangent -adj_y at EVERY evaluation of the backward
 * solve).  With f = relu(W (A y) + b) the three vector-Jacobian products are closed forms; one call, no torch graph:
 *   K     [n][H]  = ODEFunc(y)                              (func_eval)
 *   vjp_y [n][H]  = -A^T ((a (.) [K > 0]) W)                (d/dt of adj_y)
 *   vjp_W [H][H]  = -(a (.) [K > 0])^T (A y)                (d/dt of adj_params, weight part; nn.Linear's [out][in] layout)
 *   vjp_b [H]     = -column sums of a (.) [K > 0]           (bias part)
 * (d/dt of adj_t is zero: the reference ODEFunc ignores t, neural_dynamics.py:20-26.)  a = adj_y.  A_t: the operator's
 * transpose as CSR.  NDCN_F_NO_GRAPH / NDCN_F_NO_CONTROL as ndcn_rhs_f32 (NO_CONTROL: vjp_W / vjp_b are not written).
 * work: ndcn_adjoint_rhs_work_bytes() bytes, 256-byte aligned.  b nullable.                                              */
NDCN_API int ndcn_adjoint_rhs_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *y, const float *a, const float *W,
                                  const float *b, float *K, float *vjp_y, float *vjp_W, float *vjp_b, void *work, int H,
                                  uint32_t flags, void *stream);
NDCN_API int64_t ndcn_adjoint_rhs_work_bytes(int64_t n_rows, int H, uint32_t flags);

/* ODEFunc.forward PLUS the Runge-Kutta algebra that consumes its result, in one pass over the panel
 * (H = 256: inside the fused kernel's epilogue; other widths: the same result from separate kernels).
 *   rk_mode 0                  K = ODEFunc(X)                                   (= ndcn_rhs_f32)
 *   rk_mode NDCN_RK_COMBINE    also y_next = y0 + sum_{m<n_prev} h_c[m] kprev[m] + h_c[n_prev] K
 *                              (stage input of the NEXT evaluation, rk_common.py:51; K is the last term)
 *   rk_mode NDCN_RK_ERROR      also d_out = {sum ((sum_m h_c[m] k_m) / (atol + rtol max(|y0|, |X|)))^2,
 *                              non-finite count of X}: the dopri5 error record, X being y1 (rk_common.py:60,
 *                              misc.py:146-157, dopri5.py:101-102); d_ws: ndcn_reduce_ws_bytes() bytes
 *   rk_mode NDCN_RK_RK4        also y_next = the input of the next stage of the 3/8-rule RK4 step - or, for the last
 *                              stage, the step's result - in the reference's operator order (rk_common.py:72-78,
 *                              solvers.py:92); n_prev = stage index 0..3, kprev = the earlier stages, h_c[0] = dt:
 *                                0: y0 + dt K / 3            1: y0 + dt (K - k1 / 3)
 *                                2: y0 + dt (k1 - k2 + K)    3: y0 + (k1 + 3 k2 + 3 k3 + K) dt / 8
 *                              (y_next may alias y0 here: both are row-local to the epilogue)
 * y1 (nullable, NDCN_RK_ERROR only): the state whose error record is formed, indexed by ROW OF THIS LAUNCH; NULL = X, the
 *   evaluation's own input (the single-launch case).  Two callers pass it: a launch over a row block [a, b) of a shard
 *   (X stays the whole own panel because columns index it; y1 = X + a * H), and the second phase of a two-phase
 *   evaluation, whose X is the partial sum A_own X rather than the state (ndcn_amd/sharding.py).  With NDCN_F_ACCUM the
 *   record is added to d_out.
 * y_aux / h_c_aux (nullable, NDCN_RK_COMBINE only): a SECOND linear combination of the same stages, without y0,
 *   y_aux = sum_{m<n_prev} h_c_aux[m] kprev[m] + h_c_aux[n_prev] K   (left to right, products rounded on their own),
 *   written in the same pass.  dopri5 forms E = dt sum_{j<=6} c_err[j] k_j this way in the launch that produces k6, whose
 *   epilogue holds k1, k3, k4, k5 already; the error launch then runs with n_prev = 1, kprev = {E}, h_c = {1, dt c_err[7]} and
 *   reads {y0, E, y1} instead of {y0, k1, k3, k4, k5, k6, y1}: 3 P less HBM traffic per step, the same sum in the same order
 *   (rk_common.py:60; E is the exact partial sum).  NDCN_RK_ERROR therefore accepts n_prev = 1 besides dopri5's 5 in the fused kernels.
 * h_kprev / h_c are HOST arrays (n_prev <= 5 device pointers; n_prev + 1 coefficients, already dt * beta in
 * fp32).  X, K, y_next, y0 and the kprev panels must not alias each other.                              */
#define NDCN_RK_COMBINE 1
#define NDCN_RK_ERROR   2
#define NDCN_RK_RK4     3
NDCN_API int ndcn_rhs_rk_f32(const ndcn_csr *A, const float *X, const float *X_halo, int64_t n_own,
                             const float *W, const float *b, float *K, float *work, int H, uint32_t flags,
                             int rk_mode, const float *y0, const float *const *h_kprev, const float *h_c, int n_prev,
                             float *y_next, const float *y1, float *y_aux, const float *h_c_aux, float rtol, float atol,
                             double *d_out, void *d_ws, void *stream);
/* The evaluation that OPENS a dopri5 step (rk_common.py:45-51 with i = 0): K = f(X + x_add_c * x_add), and in the same pass
 * y_next = y0 + (h_c[0] * k_prev + h_c[1] * K) - i.e. ndcn_rk_combine_f32(tmp, X, {x_add}, {x_add_c}) followed by
 * ndcn_rhs_rk_f32(..., tmp, ..., NDCN_RK_COMBINE, y0, {k_prev}, h_c, 1, y_next) without the temporary: the sum is formed on
 * the neighbour rows the kernel stages (one product, one sum per element: the same bits), 3 panels of traffic less.  For the
 * operators / widths ndcn_rhs_xadd_supported() accepts (H = 256, a 2-D lattice's group-record plan, no halo panel,
 * NDCN_RK_COMBINE with n_prev = 1); NDCN_EINVAL otherwise.  In the dopri5 step X = y0, x_add = k_prev = k1.             */
NDCN_API int ndcn_rhs_xadd_supported(const ndcn_csr *A, int H, uint32_t flags, int rk_mode, int n_prev);
NDCN_API int ndcn_rhs_rk_xadd_f32(const ndcn_csr *A, const float *X, const float *x_add, float x_add_c, const float *W,
                                  const float *b, float *K, float *work, int H, uint32_t flags, const float *y0,
                                  const float *k_prev, const float *h_c, float *y_next, void *stream);

/* The two halves of odeint_adjoint's right-hand side on the fused launch (reference torchdiffeq/_impl/adjoint.py:34-59, where
 * torch.autograd.grad forms -a^T df/dy and -a^T df/dtheta at every evaluation; ndcn_amd/torchdiffeq/_impl/adjoint_fused.py):
 *   s_out  (forward half, launch over (A, W, b)): besides K and the stage algebra, S = A X is written - the weight gradient
 *          gZ^T S of the same evaluation needs it (otherwise a second SpMM);
 *   x_mask (transposed half, launch over (A^T, W^T) without bias / activation): the evaluation's input is X (.) [x_mask > 0],
 *          formed on the rows the kernel stages - X = the adjoint state, x_mask = K of the forward half, i.e. the gathered rows are
 *          gZ = a (.) [K > 0] without a panel of its own (3 panels of traffic).
 * Exactly one of the two per call; arguments otherwise as ndcn_rhs_rk_f32 (no halo panel, no y_aux).  Provided for the operators /
 * launches ndcn_rhs_adj_supported() accepts: H = 256, a 2-D lattice's group-record plan, NDCN_RK_COMBINE with n_prev = 1..4 and
 * NDCN_RK_ERROR with n_prev = 5 (the launches of a dopri5 step); NDCN_EINVAL otherwise.                                           */
NDCN_API int ndcn_rhs_adj_supported(const ndcn_csr *A, int H, uint32_t flags, int rk_mode, int n_prev);
NDCN_API int ndcn_rhs_rk_adj_f32(const ndcn_csr *A, const float *X, const float *x_mask, float *s_out, const float *W, const float *b,
                                 float *K, float *work, int H, uint32_t flags, int rk_mode, const float *y0,
                                 const float *const *h_kprev, const float *h_c, int n_prev, float *y_next, const float *y1,
                                 float rtol, float atol, double *d_out, void *d_ws, void *stream);

/* Pack rows `idx[0..n_idx)` of X into out (halo send buffers).  out[i, :] = X[idx[i], :] */
NDCN_API int ndcn_gather_rows_f32(const float *X, const int32_t *idx, int64_t n_idx, int H, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Runge-Kutta bookkeeping, one pass each (the reference issues one ATen op per term).
 * h_k: HOST array of n_k device pointers; h_c: HOST array of n_k fp32 coefficients ALREADY multiplied by
 * dt and rounded to fp32, as misc.py:25 `(scale * x)` does.  Terms are added left to right, products and
 * sums rounded separately (no FMA), exactly like `sum([...])` over tensors.
 */
/* out = y0 + sum_j c_j k_j          rk_common.py:51 via misc.py:22-25 ; dopri5.py:42 (y_mid).  y0 == NULL: sum_j c_j k_j alone
 * (the gradient combinations of a reverse sweep).                                                      */
NDCN_API int ndcn_rk_combine_f32(float *out, const float *y0, const float *const *h_k, const float *h_c, int n_k,
                        int64_t n_elem, void *stream);

/* err = sum_j c_j k_j ; tol = atol + rtol * max(|y0|, |y1|) ; r = err / tol
 * d_out[0] = sum r^2 (fp64, deterministic order), d_out[1] = number of non-finite elements of y1.
 * rk_common.py:60 + misc.py:146-157 + the finiteness assertion of the NEXT step (dopri5.py:101-102).
 * d_ws: device scratch of ndcn_reduce_ws_bytes() bytes.                                               */
NDCN_API int ndcn_rk_error_f32(const float *y0, const float *y1, const float *const *h_k, const float *h_c, int n_k,
                      float rtol, float atol, int64_t n_elem, double *d_out, void *d_ws, void *stream);

/* Reductions of ndcn_rk_error_f32 / ndcn_scaled_sumsq_f32 over at most this many elements reproduce ATen's float32 summation order
 * (torch.mean / Tensor.norm: misc.py:71-76,156 - one workgroup, serial: the reference-sized solves, whose accept / reject decisions
 * hang on the last bit); larger ones use the parallel fixed-order fp64 reduction.  Default 2^18 (environment NDCN_ATEN_NORM_MAX).
 * This call overrides the bound PROCESS-WIDE at run time (n_elem < 0: back to the default) and returns the previous override
 * (-1: none).  odeint_adjoint's fused reverse pass lowers it around its parameter-gradient vector (65 792 elements riding next
 * to 10^5..10^6-row panels: 0.6 ms serial against 10 us parallel).                                                               */
NDCN_API int64_t ndcn_set_aten_norm_max(int64_t n_elem);

/* d_out[0] = sum ((a - b) / (atol + |y| * rtol))^2  (b nullable), d_out[1] = non-finite count of a.
 * The three RMS norms of misc.py:123-138 (d0, d1, d2).                                               */
NDCN_API int ndcn_scaled_sumsq_f32(const float *a, const float *b, const float *y, float rtol, float atol,
                          int64_t n_elem, double *d_out, void *d_ws, void *stream);
NDCN_API int64_t ndcn_reduce_ws_bytes(void);

/* Dense-output fit of an accepted dopri5 step: y_mid from the 7 stage derivatives, then the quartic's
 * a, b, c, d (e aliases y0).  h_cmid = dt * DPS_C_MID rounded to fp32; dt in the state dtype.
 * dopri5.py:39-45 + interp.py:21-35.                                                                 */
NDCN_API int ndcn_dopri5_interp_fit_f32(const float *y0, const float *y1, const float *const *h_k /*7*/,
                               const float *h_cmid /*7*/, float dt, float *a, float *b, float *c, float *d,
                               int64_t n_elem, void *stream);
/* Fit and evaluate in one pass without storing the coefficients (same arithmetic as the pair above): for a step
 * that is sampled at a single tick this reads 9 panels and writes 1 instead of 13 + 6.                  */
NDCN_API int ndcn_dopri5_interp_direct_f32(const float *y0, const float *y1, const float *const *h_k /*7*/,
                                           const float *h_cmid /*7*/, float dt, const float h_xpow[5], float *out,
                                           int64_t n_elem, void *stream);
/* out = a x^4 + b x^3 + c x^2 + d x + e with h_xpow = {x^4, x^3, x^2, x, 1} formed by the caller in fp32
 * (interp.py:59-65).                                                                                  */
NDCN_API int ndcn_interp_eval_f32(const float *a, const float *b, const float *c, const float *d, const float *e,
                         const float h_xpow[5], float *out, int64_t n_elem, void *stream);

/* Fixed-grid stage algebra with the reference's operator order (fixed_grid.py:8,18-19; rk_common.py:72-78).
 *   op 0  out = y + dt * k1                          euler update (y + dt*f)
 *   op 1  out = y + k1 * dt / 2                      midpoint stage
 *   op 2  out = y + dt * k1 / 3                      rk4 stage 2
 *   op 3  out = y + dt * (k1 / -3 + k2)              rk4 stage 3
 *   op 4  out = y + dt * (k1 - k2 + k3)              rk4 stage 4
 *   op 5  out = y + (k1 + 3 k2 + 3 k3 + k4) * (dt/8) rk4 update
 * unused k pointers may be NULL.  out may alias y.                                                     */
NDCN_API int ndcn_fixed_stage_f32(int op, float *out, const float *y, const float *k1, const float *k2,
                         const float *k3, const float *k4, float dt, int64_t n_elem, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Truth dynamics of the three drivers on an N x 1 state, O(nnz).
 *   heat   : ndcn_spmm_f32 with H = 1, alpha = -k on L                    heat_dynamics.py:197-204
 *   gene   : out = -b x^f + A (x^h / (x^h + 1))                           gene_dynamics.py:201-204
 *   mutual : out = b + x(1-x/k)(x/c-1) + sum_j A_ij x_i x_j/(d + e x_j + h x_i)   (the branch that
 *            executes for N x 1, mutualistic_dynamics.py:206-216)
 */
NDCN_API int ndcn_gene_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float f, float h, void *stream);
NDCN_API int ndcn_mutual_rhs_f32(const ndcn_csr *A, const float *x, float *out, float b, float k, float c, float d,
                        float e, float h, void *stream);

/* Row-wise L1 normalisation: Y[r,:] = X[r,:] / max(sum_j |X[r,j]|, 1e-12), infinities zeroed.  Replaces
 * `row_normalization` / `RowNorm.forward` (ode_gcn.py:9-26; used by ResBlock(normalize=True) ode_gcn.py:50-57 and the
 * odeGCN input stack dgnn.py:152).  Y may alias X.                                                                   */
NDCN_API int ndcn_row_l1_normalize_f32(const float *X, float *Y, int64_t n_rows, int H, void *stream);
/* Its vector-Jacobian product (training through ResBlock(normalize=True), ode_gcn.py:50-57): GX = dL/dX given G = dL/dY
 * and the forward input X.  GX may alias G.                                                                            */
NDCN_API int ndcn_row_l1_normalize_bwd_f32(const float *G, const float *X, float *GX, int64_t n_rows, int H, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU transport (ndcn_halo_plan_create / exchange; 8e).  The path shards by node range, one rank per
 * GPU: rank r owns rows [b_r, b_{r+1}) of the operator and of every panel; per right-hand side the X rows of remote
 * column neighbours ("halo") arrive by ONE all-to-all-v - a grouped ncclSend / ncclRecv per peer over RCCL / xGMI, only
 * the referenced rows travel - and the controller's 16-byte record is all-reduced.  The reference is single-device
 * (SURVEY 8e); these calls are what its `A x` (neural_dynamics.py:29) and `torch.mean` / `norm` (misc.py:71-76,156)
 * become when the node set is split.  librccl is bound at first use (dlopen): without it these calls return NDCN_EHIP. */
typedef struct ndcn_comm ndcn_comm;
typedef struct ndcn_halo_plan ndcn_halo_plan;
/* rank 0: a 128-byte id to hand to every rank (over the caller's own channel), then every rank: create on its device */
NDCN_API int ndcn_comm_unique_id(char h_id[128]);
NDCN_API int ndcn_comm_create(const char h_id[128], int world, int rank, ndcn_comm **out);
/* or adopt the caller's ncclComm_t (passed as void*; not destroyed by ndcn_comm_destroy) */
NDCN_API int ndcn_comm_adopt(void *nccl_comm, int world, int rank, ndcn_comm **out);
/* TEST transport: the same communicator interface over POSIX shared memory, host-staged and synchronous (csrc/comm.hip) - ranks
 * that share ONE device (RCCL refuses those) or have no RCCL at all can run every multi-rank entry point of this header, the
 * device-resident sharded solver included.  `name`: the same string on every rank, unique per communicator (<= 99 characters,
 * no '/'); world <= 64.  Collective: returns when every rank has joined (120 s bound).  Never the production path.               */
NDCN_API int ndcn_comm_create_loopback(const char *name, int world, int rank, ndcn_comm **out);
NDCN_API int ndcn_comm_destroy(ndcn_comm *c);
/* d_buf[0..n) <- sum over ranks (in place, enqueued on `stream`; a no-op for world 1) */
NDCN_API int ndcn_comm_allreduce_sum_f64(ndcn_comm *c, double *d_buf, int n, void *stream);
/* h_send_counts / h_recv_counts: HOST arrays [world] of ROW counts per peer (a rank may list itself: rows routed through
 * the exchange to itself); d_send_idx: DEVICE int32 rows of the own panel to pack, grouped by peer in rank order
 * (sum of send counts entries; must outlive the plan); the halo panel receives peer 0's rows first, then peer 1's ...
 * (sum of receive counts == n_halo).  any_rank_moves_rows: the caller's GLOBAL fact (an all-reduce at plan time) - the
 * exchange is skipped only when no rank sends or receives, never on a rank-local test.                                */
NDCN_API int ndcn_halo_plan_create(ndcn_comm *c, int64_t n_halo, const int64_t *h_send_counts, const int64_t *h_recv_counts,
                                   const int32_t *d_send_idx, int any_rank_moves_rows, ndcn_halo_plan **out);
NDCN_API int ndcn_halo_plan_destroy(ndcn_halo_plan *p);
/* X_halo[n_halo, H] <- the rows of the peers' panels this rank references; d_pack: scratch of (sum of send counts) * H floats */
NDCN_API int ndcn_halo_exchange_f32(ndcn_halo_plan *p, const float *X, int H, float *d_pack, float *X_halo, void *stream);

/* How one rank's shard is evaluated by the device-resident solver (ndcn_solver_desc::shard).  desc.A is then an operator
 * over the columns [own | halo] (n_cols = n_own + n_halo).  Three forms, all with the exchange on a side stream:
 *   row split  (n_blocks > 0)  the shard's rows in up to 4 contiguous blocks; blocks with needs_halo == 0 (operator over
 *              the own columns only) run while the halo is in flight, the others after it has landed.  Lattices: one long
 *              interior block between two thin boundary bands.  desc.A is unused.
 *   two-phase  (A_own.n_rows > 0)  S = A_own X during the exchange, then desc.A = [I | A_halo] over the panels
 *              [S | X_halo] (the same fma sequence per row as one launch).  Scattered halos: small-world, power-law shards.
 *   one launch (neither)       desc.A over [X | X_halo] after the exchange.
 * n_global_rows: the node count of the whole graph (means and RMS norms of the controller are global).               */
typedef struct ndcn_shard {
    ndcn_comm      *comm;
    ndcn_halo_plan *halo;
    int64_t         n_global_rows;
    int32_t         n_blocks;
    struct { int64_t row_lo, row_hi; int32_t needs_halo; ndcn_csr A; } blocks[4];
    ndcn_csr        A_own;
    float          *X_halo;       /* nullable: where the halo rows land (n_halo x H floats).  An operator with a long-row
                                     plan wants them directly in front of its hub rows (struct ndcn_csr: hub_S == X_halo +
                                     n_halo * H); NULL: inside the solver's workspace                                    */
} ndcn_shard;

/* ------------------------------------------------------------------------------------------------
 * Device-resident integrator for the ODEFunc RHS (state, stage derivatives and dense-output coefficients
 * stay in HBM across steps; the host only reads the 16-byte error record per adaptive step).
 * Mirrors the reference's per-call solver object (odeint.py:71-72; dopri5.py:58-122; solvers.py:79-99).
 *
 * The solver lives in ONE device workspace of ndcn_solver_workspace_bytes(desc) bytes (256-byte aligned):
 * pass the caller's (e.g. a torch allocation, so repeated solves reuse cached memory) or NULL to let
 * ndcn_solver_create hipMalloc it (that synchronises; ndcn_solver_destroy frees it).  All other calls only
 * enqueue work on `stream`, except that the dopri5 controller waits for each step's 16-byte error record.
 */
typedef struct ndcn_solver ndcn_solver;

typedef struct ndcn_solver_desc {
    int      method;          /* NDCN_M_*                                                            */
    int      H;
    uint32_t rhs_flags;       /* NDCN_F_RELU | NO_GRAPH | NO_CONTROL                                  */
    int      use_graph;       /* fixed-grid methods: replay one captured hipGraph per step            */
    ndcn_csr A;
    const float *W, *b;       /* nullable under NO_CONTROL                                            */
    double   rtol, atol;      /* dopri5                                                               */
    int64_t  max_num_steps;   /* dopri5.py:61 (2^31-1 in the reference)                               */
    double   safety, ifactor, dfactor;   /* dopri5 step-size controller (dopri5.py:60,72-74; misc.py:160-170); the
                                            reference's defaults pass through a float32 tensor: (double)0.9f, 10, (double)0.2f.
                                            Values <= 0 select those defaults.                                  */
    const ndcn_shard *shard;  /* NULL: the whole graph on this device.  Otherwise this rank's shard of a node-range
                                 sharded graph: A.n_rows = own rows, A.n_cols = own + halo columns (see ndcn_shard) */
} ndcn_solver_desc;

NDCN_API int64_t ndcn_solver_workspace_bytes(const ndcn_solver_desc *desc);
NDCN_API int ndcn_solver_create(const ndcn_solver_desc *desc, void *workspace, int64_t workspace_bytes,
                                ndcn_solver **out);
NDCN_API int ndcn_solver_destroy(ndcn_solver *s);
/* Start at (t0, y0): copies y0 in; dopri5 also evaluates f0 and the initial step (dopri5.py:77-83).   */
NDCN_API int ndcn_solver_begin(ndcn_solver *s, const float *y0, double t0, void *stream);
/* As ndcn_solver_begin, without the copy (dopri5; other methods and replay mode copy as above): the solver reads the
 * initial state where it lies - odeint.py:39-41 hands y0 over and never writes it - until its second accepted step.
 * y0 must stay valid and unchanged until the next ndcn_solver_begin* / ndcn_solver_destroy on `s`, and no `out` of
 * ndcn_solver_advance / _advance_many may overlap it (refused with NDCN_EINVAL).                       */
NDCN_API int ndcn_solver_begin_borrowed(ndcn_solver *s, const float *y0, double t0, void *stream);
/* Integrate to next_t and write y(next_t) to `out` (n_rows x H).
 * fixed grid: ONE step of size next_t - t (solvers.py:89-97).
 * dopri5: adaptive steps until t1 >= next_t, at most `step_budget` of them (<= 0: unlimited), then the
 * dense-output evaluation (dopri5.py:85-92).  When the budget is exhausted first, returns 1 and leaves
 * `out` untouched; call again to continue.                                                            */
NDCN_API int ndcn_solver_advance(ndcn_solver *s, double next_t, float *out, int64_t step_budget, void *stream);
/* All of h_ticks (HOST array, increasing, the first one > the current time) in one call: out[i] = y(h_ticks[i]), n_ticks
 * panels back to back.  dopri5: ticks that fall into the same accepted step are evaluated together, reading the step's
 * panels once per <= 8 ticks instead of once per tick (the reference's drivers sample 16-120 ticks over a handful of
 * steps: heat_dynamics.py:35,123; dgnn.py:173-182); same arithmetic per element as ndcn_solver_advance.  Fixed grid:
 * one step per tick.                                                                                     */
NDCN_API int ndcn_solver_advance_many(ndcn_solver *s, const double *h_ticks, int64_t n_ticks, float *out, void *stream);
/* h_stats = {steps attempted, steps accepted, rhs evaluations, t1, dt_next, last mean_sq_error_ratio} */
NDCN_API int ndcn_solver_stats(const ndcn_solver *s, double h_stats[6]);
/* Copies up to `cap` rows {t0, dt, accepted, ratio, dt_next} of the per-attempt log; returns the count. */
NDCN_API int64_t ndcn_solver_steplog(const ndcn_solver *s, double *h_rows, int64_t cap);

/* ------------------------------------------------------------------------------------------------
 * A WHOLE fixed-grid solve in one launch, for states that fit one compute unit: FixedGridODESolver.integrate (solvers.py:79-99)
 * with Euler / midpoint / RK4-3/8 steps (fixed_grid.py:7-29, rk_common.py:72-78) over ODEFunc (neural_dynamics.py:27-36) - the
 * reference's own commands (heat_dynamics.py:20-22,33: Euler, H = 20, 400 nodes, 80-120 ticks), where a launch per step is
 * pure latency.  One workgroup keeps the stage input in LDS and the row-local panels in registers across all steps; every tick
 * is bit-identical to the per-step kernels.  ndcn_solver_advance_many takes this path by itself; the entry points are
 * public for callers that own their time loop and for the reverse sweep:
 *   h_dt      HOST array of the n_ticks step sizes, formed in the state dtype (solvers.py:81-84: t[i+1] - t[i] in fp32)
 *   out       n_ticks panels, out[i] = y(t[i+1])
 *   backward  (Euler; the drivers train by backprop through the solver, heat_dynamics.py:313-334): traj / g_out hold
 *             n_ticks + 1 panels - y(t[0]) .. y(t[n_ticks]), i.e. y0 followed by the forward's `out`, and dL/dy of each -
 *             g_y0 [n][H], g_W [H][H], g_b [H] receive the gradients (A_t: the operator's transpose as CSR).
 * supported: H <= 64 (backward: H <= 31), a plain operator (no halo), the state (backward: 4 panels) within 160 KB of LDS and
 * <= 576 * 20 / H rows; NDCN_EINVAL otherwise (ask ndcn_solve_small_supported first).                                      */
NDCN_API int ndcn_solve_small_supported(const ndcn_csr *A, int H, uint32_t flags, int method, int backward);
NDCN_API int ndcn_solve_small_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method,
                                  const float *y0, const float *h_dt, int64_t n_ticks, float *out, void *stream);
NDCN_API int ndcn_solve_small_bwd_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *W, const float *b, int H, uint32_t flags,
                                      int method, const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks,
                                      float *g_y0, float *g_W, float *g_b, void *stream);
/* Euler training with the evaluations' intermediates KEPT (the README commands' shape: H = 16 / 20, a symmetric operator whose view has
 * max_row_len and symmetric filled - ndcn_solve_small_keep_supported says whether both launches take the form): the forward launch also
 * writes, per step, S_i = A y_i and K_i = f(y_i) into `keep` (n_ticks x 2 panels); the reverse sweep reads them instead of re-forming
 * them from y_i - no gather and no Linear per tick (a third of the sweep's cycles).  Same results as the pair above.                  */
NDCN_API int ndcn_solve_small_keep_supported(const ndcn_csr *A, int H, uint32_t flags);
NDCN_API int ndcn_solve_small_keep_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, const float *y0,
                                       const float *h_dt, int64_t n_ticks, float *out, float *keep, void *stream);
NDCN_API int ndcn_solve_small_bwd_keep_f32(const ndcn_csr *A, const ndcn_csr *A_t, const float *W, const float *b, int H, uint32_t flags,
                                           const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks, const float *keep,
                                           float *g_y0, float *g_W, float *g_b, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement aid (bench.py `roofline`): when enabled, every kernel launch made by this library is
 * bracketed by HIP events recorded on the launch stream.  ndcn_prof_read drains them into
 * h_out[kind*4 + {0: launches, 1: total ms, 2: total algorithmic bytes, 3: total flops}] for
 * kind in 0..ndcn_prof_kinds()-1 = {spmm, linear, rhs_fused, combine, error, sumsq, interp_fit,
 * interp_eval, fixed_stage, gather_rows, truth_dynamics}; returns 1 if the record ring overflowed.  */
NDCN_API int ndcn_prof_enable(int on);
NDCN_API int ndcn_prof_read(double *h_out, int n_kinds);
NDCN_API int ndcn_prof_kinds(void);
/* Which kernel family the LAST fused right-hand side of this thread was dispatched to (tests: the intended native path
 * ran, not merely a correct one): bit 0 rhs_fused2, bit 1 rhs_fused3 (group-record plan), bit 2 long-row plan active,
 * bit 3 a halo panel was passed; 0 before the first call.                                                          */
#define NDCN_PATH_FUSED2 1
#define NDCN_PATH_FUSED3 2
#define NDCN_PATH_HUB    4
#define NDCN_PATH_HALO   8
#define NDCN_PATH_SWEEP 16
#define NDCN_PATH_REC   32   /* no_control: relu(A X) + the stage algebra in the epilogue of the group-record SpMM (spmm_rec.hip) */
#define NDCN_PATH_WIDE  64   /* no_control: the same in the epilogue of the row SpMM (spmm.hip: spmm_wide_kernel)                */
#define NDCN_PATH_SMALL 128  /* H <= 128: the whole ODEFunc + epilogue in one launch (rhs_small.hip)                            */
#define NDCN_PATH_EXACT32 256 /* H = 256, range guard: the weights' in-row range exceeds what the two-piece fp16 product guarantees
                               * (four or more non-zero elements more than 2^19 below their row's largest magnitude: csrc/split16.h), found when
                               * the image was packed - the launch ran with the fp32 matrix cores as its consumer (v_mfma_f32_32x32x2_f32;
                               * nn.Linear in fp32, neural_dynamics.py:33; csrc/rhs_fused2_exact.hip).  NDCN_RANGE_GUARD=0
                               * switches the guard (and the 32-byte read-back per packed image) off                              */
#define NDCN_PATH_RANGE  512 /* such weights on a launch that exists only fused (x_add / x_mask / s_out): split product, warned once */
NDCN_API int ndcn_debug_last_rhs_path(void);
/* The range guard of the H = 256 Linear (NDCN_PATH_EXACT32 above): on = 1 / 0 switches it PROCESS-WIDE at run time, on < 0 returns to the
 * default (on, unless the environment says NDCN_RANGE_GUARD=0); returns the previous state (1 / 0).  Off, every packed image takes the
 * split fp16 product whatever its range, and packing does not read back.  Images packed while the guard was off are judged when
 * they are packed next.                                                                                                            */
NDCN_API int ndcn_set_range_guard(int on);

/* ---- training through the adaptive solver: the solve that keeps its tape, and its reverse pass ---------------------------------
 * Replaces, for a plain ODEFunc on one state tensor, what the reference's drivers do by autograd through odeint
 * (heat_dynamics.py:313-334, dgnn.py:192-222; torchdiffeq/_impl/dopri5.py:58-122, rk_common.py:41-61, misc.py:84-170, interp.py:21-65):
 * ndcn_tape_dopri5_f32 integrates y' = relu(W (A y) + b) over the n_t strictly increasing ticks (out: n_t panels, the first is y0)
 * with the launches of the per-operation path (same kernels, same order: bit-identical values and accept / reject decisions) and
 * records every attempted step; ndcn_tape_backward_f32 turns g_out (n_t panels: the gradient of the trajectory) into the gradients
 * of y0, W and b - the panel operations' VJP kernels of this header plus the adjoint of the step-size controller's scalar chain
 * (dt, t0 / t1, the initial step and the interpolation abscissa carry gradient in the reference: csrc/tape.hip).  The reverse pass
 * may run any number of times over one tape (retain_graph, Jacobian rows): it only reads the record; memory it asks `alloc` for is
 * scratch of that call and may be released when the call returns (stream-ordered).
 * opts: {first_step given (0 / 1), safety, ifactor, dfactor, max_num_steps, keep S (0 / 1: evaluations that can - ndcn_rhs_adj_supported -
 * also store S = A u on the tape, one panel more per evaluation, instead of one SpMM per evaluation in the reverse pass)}.
 * At: the transposed operator (unused with NDCN_F_NO_GRAPH).
 * alloc: device memory for the tape (12 panels per attempted step; ~24 more during a reverse pass), owned by the caller; what
 * ndcn_tape_dopri5_f32 asked for is kept until ndcn_tape_destroy.  y0, W, b and the operators must stay valid and unchanged until then.
 * Errors as the solver's: NDCN_EMAXSTEPS, NDCN_EUNDERFLOW, NDCN_ENONFINITE; the tape handle is set on every path - destroy it.     */
typedef struct ndcn_tape ndcn_tape;
typedef void *(*ndcn_alloc_fn)(void *ctx, int64_t bytes);
NDCN_API int ndcn_tape_dopri5_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags,
                                  const float *y0, const double *ticks, int64_t n_t, double rtol, double atol, const double *opts,
                                  float *out, ndcn_alloc_fn alloc, void *alloc_ctx, ndcn_tape **tape, void *stream);
NDCN_API int ndcn_tape_backward_f32(ndcn_tape *tape, const float *g_out, float *g_y0, float *g_W, float *g_b, void *stream);
NDCN_API int64_t ndcn_tape_steplog(const ndcn_tape *tape, double *rows, int64_t cap);   /* 5 doubles per attempt, as ndcn_solver_steplog */
NDCN_API int64_t ndcn_tape_nfe(const ndcn_tape *tape);
NDCN_API void ndcn_tape_destroy(ndcn_tape *tape);
/* The fixed-grid methods (solvers.py:79-99 with fixed_grid.py:7-29, rk_common.py:72-78; method: NDCN_M_EULER / _MIDPOINT / _RK4) the same
 * way, without a tape object - the reverse pass re-forms the stages of a step from the stored trajectory:
 * ndcn_fixed_grid_train_f32 writes the n_ticks + 1 states (out[0] = y0) for the step sizes h_dt (the grid's differences in the state
 * dtype, solvers.py:81), ndcn_fixed_grid_backward_f32 maps g_out (n_ticks + 1 panels) to the gradients of y0, W and b.  Any size; the
 * launches of _impl/odeint.py::_FixedGridSolve.  alloc: scratch (<= 22 panels), may be released when the call returns (stream-ordered). */
NDCN_API int ndcn_fixed_grid_train_f32(const ndcn_csr *A, const float *W, const float *b, int H, uint32_t flags, int method,
                                       const float *y0, const float *h_dt, int64_t n_ticks, float *out, ndcn_alloc_fn alloc,
                                       void *alloc_ctx, void *stream);
NDCN_API int ndcn_fixed_grid_backward_f32(const ndcn_csr *A, const ndcn_csr *At, const float *W, const float *b, int H, uint32_t flags,
                                          int method, const float *traj, const float *g_out, const float *h_dt, int64_t n_ticks,
                                          float *g_y0, float *g_W, float *g_b, ndcn_alloc_fn alloc, void *alloc_ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NDCN_HIP_H */
