#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on the metric's own case.

Metric   : node-states/sec = N x T_steps / wall, T_steps = dopri5 steps ATTEMPTED (accepted + rejected),
           state already resident in HBM.
Workload : NDCN ODEFunc relu(W (A X) + b) on the 1M-node 8-neighbour grid (1000 x 1000), normalised-Laplacian
           operator (nnz 8 988 004), H = 256, dopri5 rtol .01 / atol .001 on t in [0, 5], fp32.
           One bench "step" = one attempted adaptive step of the device-resident solver (6 RHS evaluations +
           stage algebra + error norm + the 16-byte controller read-back).  When a solve reaches t = 5 it is
           restarted from x0 inside the timed region (its f0 / initial-step evaluations are paid for too).
N > 1    : weak scaling - the grid grows to (1000 N) x 1000, node-range sharded, one rank per GPU, halo rows
           exchanged over RCCL before every RHS (ndcn_amd/sharding.py).
--config : M (default, the judged line) or one of BASELINE.json's other single-GPU configurations as parity / measurement
           cases with the same JSON contract: C2 (100k-node G(n,p), RK4 on linspace(0,5,100)), C3 (1M-node
           Barabasi-Albert m=5, dopri5), C5 (dgnn hot path: Pubmed topology, no_control, dopri5 rtol=atol=.1, 16 ticks).

Prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on the
launch stream during a second, instrumented pass over the same K steps; `traffic` from the committed PMC summary of THIS
configuration or null) and `cpu_baseline` (the CPU oracle timed on a bounded sample of this configuration, rank 0, N = 1 only).
`--gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16 / _bf16 dense peak (no sparsity)
SPLIT_PRODUCTS = 3             # the fused H = 256 kernels form W S from two fp16 pieces per operand: 3 fp16 MFMA products per fp32 product (split16.h)


def in_family(kind, kernel_name):
    """Does a rocprofv3 kernel name belong to a ProfScope kind of the library?  'rhs_fused' = every launch that carries the
    right-hand side WITH an epilogue or the Linear: the fused MFMA kernels, rhs_small, and the SpMM kernels in a non-plain
    mode (their last template argument: the no_control RHS + RK epilogue)."""
    import re
    if kind == 'rhs_fused':
        if 'rhs_fused' in kernel_name or 'rhs_small' in kernel_name:
            return True
        m = re.search(r'spmm_(rec|wide)_kernel<([^>]*)>', kernel_name)
        return bool(m) and m.group(2).split(',')[-1].strip() != '0'
    if kind == 'spmm':
        m = re.search(r'spmm_(rec|wide)_kernel<([^>]*)>', kernel_name)
        return 'spmm_csr' in kernel_name or 'spmm_sweep' in kernel_name or (bool(m) and m.group(2).split(',')[-1].strip() == '0')
    return {'combine': 'combine_kernel', 'linear': 'linear_', 'error': 'rk_error'}.get(kind, kind) in kernel_name


def pmc_traffic(kind, cfg):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary OF THIS CONFIGURATION
    (profiles/*_traffic_pmc_<cfg>.json, produced by tools/gpu.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  Launch-weighted mean over the family's kernels; (None, None)
    when no summary of this configuration is committed - a number measured on another workload is not this line's traffic."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic_pmc_%s.json' % cfg)))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))['kernels']
    except Exception:
        return None, None
    num = den = 0
    biggest = 0
    for name, v in d.items():
        if in_family(kind, name) and 'finish' not in name:
            num += v['hbm_bytes_per_launch'] * v['launches']
            den += v['launches']
            biggest = max(biggest, v['hbm_bytes_per_launch'])
    if not den:
        return None, None
    if kind == 'rhs_fused':
        # operators with the column-sweep plan evaluate the right-hand side in TWO launches (spmm_sweep_kernel: S = A X, then the
        # fused kernel on the identity operator over S; one ProfScope spans both): the evaluation's traffic is the sum
        sweep = [v['hbm_bytes_per_launch'] for name, v in d.items() if 'spmm_sweep_kernel' in name]
        if sweep:
            num += sweep[0] * den
    # the standalone SpMM of an operator with a long-row plan is one main launch + small hub-segment launches of the same
    # kernel family: its traffic is the main launch's, not the launch-weighted mean (which read 1.10 x for C3 in round 3
    # where the main launch moves 4.9 x)
    return (int(biggest if kind == 'spmm' else num / den), os.path.relpath(files[-1], ROOT))


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)        # ~120 dopri5 RHS evaluations (north_star)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--side', type=int, default=1000, help='grid side per GPU (N = side^2 nodes per GPU)')
    p.add_argument('--hidden', type=int, default=256)
    p.add_argument('--T', type=float, default=5.0)
    p.add_argument('--rtol', type=float, default=0.01)
    p.add_argument('--atol', type=float, default=0.001)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-threads', type=int, default=32, help='host threads of the CPU-baseline leg')
    p.add_argument('--cpu-nodes', type=int, default=0, help='nodes of the bounded CPU-baseline sample (0: per-configuration default)')
    p.add_argument('--no-at-scale', action='store_true', help='skip cpu_baseline.at_scale (one oracle RHS + one solver step at ~10^5 nodes)')
    p.add_argument('--no-profile-pass', action='store_true')
    p.add_argument('--sharded', action='store_true', help='force the multi-GPU code path (works with 1 rank)')
    p.add_argument('--sharded-impl', default=os.environ.get('NDCN_SHARDED_IMPL', 'device'), choices=['device', 'python'],
                   help='N > 1: the sharded device-resident solver behind the C ABI (default: the form DESIGN section 6 projects - the '
                        'library\'s RCCL communicator, halo exchange inside the step loop; passes with world = 8 over the loopback '
                        'transport, tests/test_gpu_eight_ranks.py; every precondition that can fail on one rank is all-reduced and '
                        'the run falls back COLLECTIVELY) or the Python-stepped path over torch.distributed')
    p.add_argument('--config', default='M', choices=['M', 'C2', 'C3', 'C4', 'C5'], help='workload (default M = the metric\'s own case)')
    p.add_argument('--no-control', action='store_true', help='config M with ODEFunc(no_control=True): relu(A X), the pure HBM right-hand side (neural_dynamics.py:32)')
    p.add_argument('--layout', default=None, choices=['degree', 'community'], help='C2 / C3: node re-labelling (--layout of the drivers)')
    p.add_argument('--method', default='dopri5', choices=['dopri5', 'euler', 'rk4'],
                   help='config M: the integrator (dopri5 = the judged line; euler - the reference drivers\' default, heat_dynamics.py:20-22 - '
                        'and rk4 step along t = linspace(0, T, 100), heat_dynamics.py:35,123)')
    p.add_argument('--cpu-runs', type=int, default=5, help='timed solves of the CPU-baseline leg (after two warm-ups; BASELINE.md section 3)')
    return p.parse_args()


class SingleGpuRunner:
    """Counts steps of the device-resident solver - attempted dopri5 steps, or fixed-grid steps along `ticks` -
    restarting from x0 when the end of the time grid is reached (the restart is inside the timed region)."""

    def __init__(self, f, x0, T, rtol, atol, method='dopri5', ticks=None, use_graph=False):
        from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
        self.solver = DeviceSolver(f, x0.shape[0], method, rtol, atol, use_graph=use_graph)
        self.x0, self.T, self.method = x0, T, method
        self.ticks = [float(v) for v in (ticks if ticks is not None else [0.0, T])]
        self.out = torch.empty_like(x0)
        # fixed grid: odeint() gives every tick a panel of its own (the Euler update then rides in the RHS epilogue: the
        # step's result is written where the next step reads it); two panels in turn reproduce that without holding 99 GB
        self.out2 = torch.empty_like(x0) if method != 'dopri5' else None
        self.solver.begin(x0, self.ticks[0], borrow=True)      # as odeint() hands the initial state over
        self.pos = 1                                     # next tick to reach
        self.solve_steps = 0                             # attempted steps of one whole solve (known after the first)
        self.traj = torch.empty((len(self.ticks) - 1,) + tuple(x0.shape), device=x0.device) if len(self.ticks) > 2 and method == 'dopri5' else None
        self.restarts = 0
        self.nfe_done = 0

    def run_steps(self, k):
        done = 0
        while done < k:
            if (self.method == 'dopri5' and len(self.ticks) > 2 and self.pos == 1 and self.solve_steps
                    and k - done >= self.solve_steps):
                # a whole solve in ONE library call, as odeint() does it (ticks of a step are evaluated together); its
                # step count is known from the previous solve of the same initial value
                self.solver.advance_many(self.ticks[1:], self.traj)
                done += int(self.solver.stats()['steps'])
                reached_end = True
            else:
                before = self.solver.stats()['steps']
                dst = self.out if (self.out2 is None or self.pos % 2) else self.out2
                reached = self.solver.advance(self.ticks[self.pos], dst, step_budget=k - done)
                done += int(self.solver.stats()['steps'] - before)
                reached_end = False
                if reached:
                    self.pos += 1
                    reached_end = self.pos == len(self.ticks)
            if reached_end:
                self.solve_steps = int(self.solver.stats()['steps'])
                self.nfe_done += int(self.solver.stats()['nfe'])
                self.solver.begin(self.x0, self.ticks[0], borrow=True)        # resets the solver's own counters
                self.pos = 1
                self.restarts += 1
        return done

    def nfe(self):
        return self.nfe_done + int(self.solver.stats()['nfe'])


# bounded CPU samples per configuration: {nodes, ticks} such that 2 warm-ups + 5 timed solves stay within ~10-30 s of CPU work
CPU_SAMPLE = {'M': 128 * 128, 'NC': 128 * 128, 'C2': 10000, 'C3': 16000, 'C4': 16000, 'C5': 0}   # (C2: >= 8192 nodes, so that the parity block's HIP solve takes the column sweep like the full configuration)


FIXED_GRID_METHOD = None         # set by main(): config M with --method euler / rk4


def cpu_workload(cfg, H, n, T):
    """(operator as scipy CSR, no_control, method, ticks, description) of configuration cfg at n nodes - the same
    generators, seeds and solver settings as build_workload."""
    from ndcn_amd import graphs
    if cfg in ('M', 'NC'):
        side = int(round(n ** 0.5))
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
        if FIXED_GRID_METHOD:
            return (L, cfg == 'NC', FIXED_GRID_METHOD, torch.linspace(0., T, 100)[:21].tolist(),
                    '%dx%d grid, the first 20 of the 99 %s steps' % (side, side, FIXED_GRID_METHOD))
        return L, cfg == 'NC', 'dopri5', [0., T], '%dx%d grid' % (side, side)
    if cfg == 'C2':
        return (graphs.normalized_laplacian(graphs.make_graph('random', n, seed=0)), False, 'rk4',
                torch.linspace(0., 5., 100)[:21].tolist(), 'G(n,p) n=%d mean degree 39.9, the first 20 of the 99 RK4 steps' % n)
    if cfg == 'C3':
        return graphs.normalized_laplacian(graphs.make_graph('power_law', n, seed=0)), False, 'dopri5', [0., T], 'Barabasi-Albert n=%d m=5' % n
    if cfg == 'C4':
        return graphs.normalized_laplacian(graphs.make_graph('small_world', n, seed=0)), False, 'dopri5', [0., T], 'small world n=%d k=5 p=0.5' % n
    import scipy.sparse as sp
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'operators_pubmed.npz')))
    m = sp.csr_matrix((g['alpha00_data'], g['alpha00_indices'], g['alpha00_indptr']), shape=(int(g['n']), int(g['n'])))
    return m, True, 'dopri5', torch.linspace(0., 1.2, 16).tolist(), 'Pubmed topology, FULL size (%d nodes), 16 ticks' % int(g['n'])


AT_SCALE_NODES = {'M': 316 * 316, 'NC': 316 * 316, 'C2': 100000, 'C3': 100000, 'C4': 100000, 'C5': 0}


def _oracle_case(cfg, H, n, T, rtol, atol):
    """(oracle ODEFunc, x0, ticks, method, rtol, atol, description, scipy operator, Linear) of configuration cfg at n nodes."""
    from oracle import ndcn_oracle as orc
    L, no_control, method, ticks, what = cpu_workload(cfg, H, n, T)
    if cfg == 'C5':
        rtol, atol = .1, .1
    A = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
    torch.manual_seed(0)
    lin = torch.nn.Linear(H, H)
    f = orc.OracleODEFunc(A, lin.weight.detach(), lin.bias.detach(), no_control=no_control)
    x0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(0))
    return f, x0, torch.tensor(ticks), method, rtol, atol, what, L, lin


def cpu_at_scale(cfg, H, T, rtol, atol):
    """BASELINE.md section 3's fall-back for sizes the reference-style solver cannot finish in minutes: ONE timed right-hand
    side and ONE timed solver step of the oracle at the configuration's full node count - or at 10^5 nodes where the full
    count needs ~40 GB of temporaries and minutes per step (M, C3, C4) - extrapolated to node-states/s as N / step time.
    dopri5: the timed unit is a solve to a tick inside the first step, i.e. the initial-step selection (2 evaluations and
    three norms) + one attempted step (6 evaluations, stage sums, error ratio) + the dense-output evaluation; rk4: the
    first grid step."""
    from oracle import ndcn_oracle as orc
    n = AT_SCALE_NODES[cfg]
    if not n:
        return None
    f, x0, tt, method, rtol, atol, what, L, _ = _oracle_case(cfg, H, n, T, rtol, atol)
    n = L.shape[0]
    f(0.0, x0)                                                       # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    for _ in range(3):
        f(0.0, x0)
    rhs_s = (time.perf_counter() - t0) / 3
    one = torch.tensor([0., 1e-4]) if method == 'dopri5' else tt[:2]
    log = []
    nfe0 = f.nfe
    t0 = time.perf_counter()
    orc.odeint(f, x0, one, rtol=rtol, atol=atol, method=method, step_log=log if method == 'dopri5' else None)
    step_s = time.perf_counter() - t0
    steps = len([r for r in log if r[0] != 'nfe']) if method == 'dopri5' else 1
    return {'nodes': n, 'what': what, 'rhs_s': round(rhs_s, 4), 'node_rhs_per_s': round(n / rhs_s, 1),
            'one_step_solve_s': round(step_s, 3), 'steps_in_it': steps, 'rhs_evals_in_it': f.nfe - nfe0,
            'share_of_time_outside_the_rhs': round(1.0 - (f.nfe - nfe0) * rhs_s / step_s, 3),
            'value_extrapolated': round(n * steps / step_s, 1), 'unit': 'node-states/s',
            'note': 'single samples (one solve after one warm-up evaluation); the reference-style solver at N >= 10^6 needs '
                    '~40 panels of temporaries per step'}


def gpu_parity(f_or, x0, tt, method, rtol, atol, ref, ref_log, dev):
    """The HIP path on the CPU leg's own sample (same operator, weights, x0, ticks, tolerances): trajectory L1 / max-abs
    against the oracle's and equality of the dopri5 accept / reject sequence (SURVEY 8d "parity gate"; north_star: L1 < 1e-4)."""
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    A = f_or.A.coalesce()
    idx = A.indices()
    op = CsrOperator.from_coo(idx[0].to(dev), idx[1].to(dev), A.values().to(dev), A.shape)
    H = x0.shape[1]
    g = ODEFunc(H, op, no_control=f_or.no_control).to(dev).eval()
    g.load_state_dict({'wt.weight': f_or.W, 'wt.bias': f_or.b})
    log = []
    with torch.no_grad():
        y = ode.odeint(g, x0.to(dev), tt.to(dev), rtol=rtol, atol=atol, method=method, step_log=log if method == 'dopri5' else None)
    err = (y.cpu() - ref).abs()
    from ndcn_amd import _lib
    bits = int(_lib.load().ndcn_debug_last_rhs_path())
    path = '+'.join(n_ for b_, n_ in ((_lib.PATH_FUSED2, 'fused2'), (_lib.PATH_FUSED3, 'fused3'), (_lib.PATH_HUB, 'long-row plan'),
                                      (_lib.PATH_HALO, 'halo'), (_lib.PATH_SWEEP, 'column sweep'),
                                      (_lib.PATH_REC, 'group-record SpMM with the RK epilogue (spmm_rec)'),
                                      (_lib.PATH_WIDE, 'row SpMM with the RK epilogue (spmm_wide)'),
                                      (_lib.PATH_SMALL, 'one-launch narrow ODEFunc (rhs_small)')) if bits & b_) or 'composed'
    mine = [bool(r[2]) for r in log if r[0] != 'nfe']
    theirs = [bool(r[2]) for r in ref_log if r[0] != 'nfe']
    return {'l1': float(err.mean()), 'max_abs': float(err.max()), 'ref_max_abs': float(ref.abs().max()),
            'steps_equal': (mine == theirs) if method == 'dopri5' else None,
            'attempts': len(mine) if method == 'dopri5' else len(tt) - 1, 'l1_bound': 1e-4, 'rhs_kernels': path}


def cpu_baseline(cfg, H, T, rtol, atol, threads, runs=5, nodes=0, dev=None, at_scale=True):
    """The CPU oracle (torch-CPU restatement of the reference path: torch.sparse.mm on COO + F.linear + the restated
    solver loops, checked against fixtures of the reference itself) on a BOUNDED sample of this configuration's workload:
    the same generators, seeds, H, tolerances and time grid at a node count that one solve finishes in seconds (C5: the
    full workload).  Protocol: two warm-up solves, `runs` >= 5 timed solves, median.
    node-states/s is a per-node rate measured at the sample's N; `at_scale` (cpu_at_scale) is the single-step figure at
    the full node count (or 10^5 nodes), where the reference-style solver falls out of the caches - BASELINE.md measured
    4.8 k node-states/s at N = 10^5 on 8 cores.  Returns (cpu_baseline, parity): parity = the HIP path on the very sample."""
    from oracle import ndcn_oracle as orc
    # the reference's op-per-term solver issues ~300 small tensor ops per step: beyond a few dozen threads the
    # fork/join cost of each op outweighs the work, so the leg uses a bounded thread count and says which
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    n = nodes or CPU_SAMPLE[cfg]
    f, x0, tt, method, rtol, atol, what, L, _ = _oracle_case(cfg, H, n, T, rtol, atol)
    n = L.shape[0]
    times, steps, ref, log = [], 0, None, []
    for r in range(2 + max(runs, 5)):
        log = []
        nfe0 = f.nfe
        t0 = time.perf_counter()
        ref = orc.odeint(f, x0, tt, rtol=rtol, atol=atol, method=method, step_log=log if method == 'dopri5' else None)
        dt = time.perf_counter() - t0
        if r >= 2:
            times.append(dt)
        nfe = f.nfe - nfe0
        steps = len([row for row in log if row[0] != 'nfe']) if method == 'dopri5' else len(tt) - 1
    med = float(np.median(times))
    sample = ('2 warm-ups + %d timed %s solves on %s (N=%d, H=%d, same generators / seeds / tolerances / time grid as '
              'the GPU run): %d steps, %d RHS evals per solve, median %.2f s (all: %s); per-node rate at the sample\'s N - see '
              'at_scale for the full node count' % (len(times), method, what, n, H, steps, nfe, med,
                                                    ', '.join('%.2f' % v for v in times)))
    parity = None
    if dev is not None:
        parity = gpu_parity(f, x0, tt, method, rtol, atol, ref, log, dev)
        parity['sample'] = '%s, N=%d, H=%d, %s, %d ticks: the CPU leg\'s own sample' % (what, n, H, method, len(tt))
    base = {'value': n * steps / med, 'unit': 'node-states/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'sample': sample}
    if at_scale:
        sc = cpu_at_scale(cfg, H, T, rtol, atol)
        if sc:
            # the figure that LEADS is the one measured where the reference-style solver has fallen out of the caches (the
            # configuration's full node count, or 10^5 nodes): the small sample above overstates the CPU ~4.6 x (round-4 review)
            base = {'value': sc['value_extrapolated'], 'unit': 'node-states/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                    'sample': 'ONE oracle solve to a tick inside the first step at N=%d (%s): %d attempted step(s), %d RHS evals, %.2f s; '
                              'one RHS alone %.3f s (BASELINE.md section 3 fall-back: the op-per-term solver needs ~40 panels of temporaries '
                              'per step and minutes per solve at this size)' % (sc['nodes'], sc['what'], sc['steps_in_it'], sc['rhs_evals_in_it'],
                                                                                  sc['one_step_solve_s'], sc['rhs_s']),
                    'at_scale': sc,
                    'cache_resident_sample': {'value': n * steps / med, 'unit': 'node-states/s', 'sample': sample}}
    return base, parity


C4_NODES_PER_GPU = int(os.environ.get('NDCN_C4_NODES', '500000'))    # BASELINE config 4: a 4M-node small world over 8 GPUs (override: tests)


def build_workload(args, dev):
    """(ODEFunc, x0, runner kwargs, description) of a single-GPU configuration (BASELINE.json configs; SURVEY 8d inputs)."""
    from ndcn_amd import graphs, CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    H = args.hidden
    torch.manual_seed(0)
    if args.config == 'M':
        S = args.side
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(S))
        A = graphs.to_device(L, dev)
        f = ODEFunc(H, A, no_control=args.no_control).to(dev).eval()
        name = ('NDCN ODEFunc(no_control) relu(AX)' if args.no_control else 'NDCN ODEFunc relu(W(AX)+b)')
        if args.method == 'dopri5':
            kw = dict(T=args.T, rtol=args.rtol, atol=args.atol, method='dopri5')
            what = (name + ', %dx%d 8-neighbour grid per GPU (N=%d nodes total), normalised-Laplacian CSR '
                    'nnz=%d per GPU, H=%d, dopri5 rtol=%g atol=%g t in [0,%g]' % (S, S, S * S, L.nnz, H, args.rtol, args.atol, args.T))
            step = 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller)'
        else:
            kw = dict(T=args.T, rtol=args.rtol, atol=args.atol, method=args.method, ticks=torch.linspace(0., args.T, 100).tolist())
            what = (name + ', %dx%d 8-neighbour grid per GPU (N=%d nodes total), normalised-Laplacian CSR nnz=%d per GPU, H=%d, '
                    'fixed-step %s on t = linspace(0,%g,100) (heat_dynamics.py:35,123)'
                    % (S, S, S * S, L.nnz, H, {'euler': 'Euler (the reference drivers\' default method)', 'rk4': 'RK4 (3/8 rule)'}[args.method], args.T))
            step = {'euler': 'one Euler step (1 RHS eval with the update in its epilogue)',
                    'rk4': 'one RK4 step (4 RHS evals with the stage algebra in their epilogues)'}[args.method]
    elif args.config == 'C2':
        G = graphs.make_graph('random', 100000, seed=0, layout=args.layout)
        L = graphs.normalized_laplacian(G)
        A = graphs.to_device(L, dev)
        f = ODEFunc(H, A).to(dev).eval()
        kw = dict(T=5.0, rtol=args.rtol, atol=args.atol, method='rk4', ticks=torch.linspace(0., 5., 100).tolist())
        what = ('C2: NDCN ODEFunc relu(W(AX)+b), G(n,p) n=100000 mean degree 39.9 (heat_dynamics.py:89 density kept), layout %s, '
                'normalised-Laplacian CSR nnz=%d, H=%d, fixed-step RK4 (3/8 rule) on linspace(0,5,100)' % (args.layout, L.nnz, H))
        step = 'one RK4 step (4 RHS evals with the stage algebra in their epilogues)'
    elif args.config == 'C3':
        G = graphs.make_graph('power_law', 1000000, seed=0, layout=args.layout)
        L = graphs.normalized_laplacian(G)
        A = graphs.to_device(L, dev)
        f = ODEFunc(H, A).to(dev).eval()
        kw = dict(T=args.T, rtol=args.rtol, atol=args.atol, method='dopri5')
        what = ('C3: NDCN ODEFunc relu(W(AX)+b), Barabasi-Albert n=1000000 m=5 (max degree %d), layout %s, normalised-Laplacian '
                'CSR nnz=%d, H=%d, dopri5 rtol=%g atol=%g t in [0,%g]'
                % (int(np.diff(L.indptr).max()), args.layout, L.nnz, H, args.rtol, args.atol, args.T))
        step = 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller)'
    elif args.config == 'C4':
        n = C4_NODES_PER_GPU
        G = graphs.make_graph('small_world', n, seed=0, layout=args.layout)
        L = graphs.normalized_laplacian(G)
        A = graphs.to_device(L, dev)
        f = ODEFunc(H, A).to(dev).eval()
        kw = dict(T=args.T, rtol=args.rtol, atol=args.atol, method='dopri5')
        what = ('C4 (one GPU\'s share of the 8-GPU configuration): NDCN ODEFunc relu(W(AX)+b), Newman-Watts-Strogatz small world '
                'n=%d k=5 p=0.5 (gene_dynamics.py:103), layout %s, normalised-Laplacian CSR nnz=%d, H=%d, dopri5 rtol=%g atol=%g '
                't in [0,%g]' % (n, args.layout, L.nnz, H, args.rtol, args.atol, args.T))
        step = 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller)'
    else:
        g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'operators_pubmed.npz')))
        n = int(g['n'])
        A = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
        L = A
        f = ODEFunc(H, A, no_control=True).to(dev).eval()
        kw = dict(T=1.2, rtol=.1, atol=.1, method='dopri5', ticks=torch.linspace(0., 1.2, 16).tolist())
        what = ('C5: dgnn.py differential_gcn hot path - ODEFunc(no_control) relu(AX) on the Pubmed topology (%d nodes, nnz=%d, '
                'operator of the committed fixture), H=%d, synthetic features, dopri5 rtol=atol=0.1, 16 ticks on [0,1.2]'
                % (n, A.nnz, H))
        step = 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller); dense output at 15 ticks per solve'
    n = A.shape[0]
    x0 = torch.rand(n, H, generator=torch.Generator().manual_seed(0)).to(dev)
    nnz = int(L.nnz)
    return f, A, x0, kw, nnz, what + ', state X~U(0,1) seed 0, nn.Linear default init seed 0', step


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with one rank per GPU (what
    the driver does itself for N > 1); rank 0's JSON line passes through on stdout, the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    global FIXED_GRID_METHOD
    if args.method != 'dopri5':
        assert args.config == 'M' and args.gpus == 1, '--method applies to the single-GPU config M'
        FIXED_GRID_METHOD = args.method
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test hook (tests/test_gpu_odeint.py): several ranks on ONE device over gloo, to exercise the N > 1 code path on a
    # 1-GPU box; RCCL refuses two ranks per device, production is always nccl with one rank per GPU
    backend = os.environ.get('NDCN_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    assert torch.cuda.is_available(), 'bench.py needs a ROCm device (there is no CPU path)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_PORT', '29655')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, '--gpus %d but WORLD_SIZE %d' % (args.gpus, world)

    from ndcn_amd import _lib, graphs, device_info
    from ndcn_amd.neural_dynamics import ODEFunc
    lib = _lib.load()

    S, H = args.side, args.hidden
    spmm_line = None
    if world == 1 and not args.sharded:
        f, A_op, x0, kw, nnz, what, step_desc = build_workload(args, dev)
        n_local = x0.shape[0]
        runner = SingleGpuRunner(f, x0, kw['T'], kw['rtol'], kw['atol'], method=kw['method'], ticks=kw.get('ticks'))
        if H == 256 and not f.no_graph:
            # the north-star's own kernel figure: the standalone CSR SpMM of this workload's operator (HIP events)
            from ndcn_amd import hip as _hip
            Y = torch.empty_like(x0)
            for _ in range(3):
                _hip.spmm(A_op, x0, out=Y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _hip.spmm(A_op, x0, out=Y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            byt = graphs.spmm_bytes(n_local, nnz, H)
            spmm_line = {'avg_ms': round(ms, 4), 'alg_bytes': byt, 'GBps': round(byt / ms / 1e6, 1),
                         'frac_of_hbm_peak': round(byt / ms / 1e6 / HBM_PEAK_GBS, 4),
                         'plan': None if A_op.rec is None else 'group-record %d rows / %d columns' % (A_op.rec['rows'], A_op.rec['cap']),
                         'long_row_plan': None if getattr(A_op, 'hub', None) is None else '%d hub rows' % A_op.hub['n'],
                         'sweep_plan': None if getattr(A_op, 'sweep', None) is None else
                         'column sweep: %d pass(es), %d rows per wave, blocks of %d columns, window %d' % (
                             A_op.sweep['passes'], A_op.sweep['rows_per_wave'], 1 << A_op.sweep['logb'], A_op.sweep['window'])}
            cfg_spmm = 'NC' if (args.config == 'M' and args.no_control) else args.config
            tr, src = pmc_traffic('spmm', cfg_spmm)
            if tr:
                spmm_line.update({'traffic': tr, 'traffic_source': src, 'traffic_over_algorithmic': round(tr / byt, 3)})
            del Y
    else:
        assert args.config in ('M', 'C4'), 'the sharded path runs the metric\'s grid or config 4\'s small world'
        torch.manual_seed(0)
        f = ODEFunc(H, None).to(dev).eval()                      # nn.Linear default init, seed 0
        from ndcn_amd import sharding
        device_impl = args.sharded_impl == 'device' and backend == 'nccl'

        def make_runner(block, bounds):
            """The Python-stepped sharded path over torch.distributed (default) or the sharded device-resident solver
            behind the C ABI.  Every step towards the latter that can fail on ONE rank is a rank-local test whose outcome is
            all-reduced BEFORE any rank enters a collective of that path: no rank may wait in a collective its peers never
            enter.  (1) the plan: a collective of the torch group both paths need - every rank builds it; (2) librccl binds
            and draws an id on every rank (local) -> all-reduce; (3) DeviceShard: its broadcast is reached by every rank
            whatever happened on rank 0 (sharding.py), ncclCommInitRank is entered by all or none; (4) the outcome of
            (3) and of the solver set-up is all-reduced again before the first step."""
            plan = sharding.bench_plan(block, bounds, rank, dev)
            r, err = None, None
            if device_impl:
                ok = torch.tensor([1 if sharding.rccl_usable() else 0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    try:
                        r = sharding.ShardedDeviceBench(f, block, bounds, rank, dev, args.T, args.rtol, args.atol, plan=plan)
                    except Exception as e:                              # noqa: BLE001 - reported below, then a collective decision
                        err = e
                    ok = torch.tensor([0 if r is None else 1], device=dev)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                else:
                    err = 'librccl not usable on some rank'
                if int(ok.item()) == 0:
                    print('[bench] rank %d: device-resident sharded solver unavailable (%r) - Python-stepped path' % (rank, err),
                          file=sys.stderr, flush=True)
                    r = None
            if r is None:
                r = sharding.ShardedBench(f, block, bounds, rank, dev, args.T, args.rtol, args.atol, plan=plan)
            return r

        if args.config == 'M':
            n_local = S * S
            runner = make_runner(*sharding.grid_row_block(S, world, rank))
            nnz = runner.local_nnz
            what = ('NDCN ODEFunc relu(W(AX)+b), %dx%d 8-neighbour grid per GPU (N=%d nodes total), normalised-Laplacian CSR nnz=%d '
                    'per GPU, H=%d, dopri5 rtol=%g atol=%g t in [0,%g], state X~U(0,1) seed = rank, nn.Linear default init seed 0'
                    % (S, S, n_local * world, nnz, H, args.rtol, args.atol, args.T))
        else:
            # weak scaling of config 4: a (500k x world)-node small world, node-range sharded; every rank generates the
            # same graph (O(n) generator, seed 0) and keeps its own rows of the normalised Laplacian
            from ndcn_amd.sharding import even_bounds
            n_glob = C4_NODES_PER_GPU * world
            L = graphs.normalized_laplacian(graphs.make_graph('small_world', n_glob, seed=0)).tocsr()
            bounds = even_bounds(n_glob, world)
            n_local = int(bounds[rank + 1] - bounds[rank])
            runner = make_runner(L[bounds[rank]:bounds[rank + 1]], bounds)
            del L
            nnz = runner.local_nnz
            what = ('C4: NDCN ODEFunc relu(W(AX)+b), Newman-Watts-Strogatz small world k=5 p=0.5, %d nodes per GPU (N=%d nodes '
                    'total), node-range sharded, halo rows by all-to-all-v per RHS (%d halo rows on this rank), normalised-Laplacian '
                    'CSR nnz=%d on rank 0, H=%d, dopri5 rtol=%g atol=%g t in [0,%g], state X~U(0,1) seed = rank, nn.Linear default '
                    'init seed 0' % (n_local, n_glob, runner.plan.n_halo, nnz, H, args.rtol, args.atol, args.T))
        step_desc = 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller)'

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then EXACTLY K steps between barriers
    with torch.no_grad():
        runner.run_steps(args.warmup)
        nfe0 = runner.nfe()
        barrier()
        t0 = time.perf_counter()
        done = runner.run_steps(args.steps)
        barrier()
        wall = time.perf_counter() - t0
    assert done == args.steps
    nfe = runner.nfe() - nfe0
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    n_total = n_local * world
    value = n_total * args.steps / wall

    # ---- instrumented pass over the same K steps: HIP events around every kernel launch
    roofline, breakdown = None, {}
    if not args.no_profile_pass:
        nk = lib.ndcn_prof_kinds()
        buf = (_lib.ctypes.c_double * (4 * nk))()
        lib.ndcn_prof_enable(1)
        lib.ndcn_prof_read(buf, nk)                          # drain
        with torch.no_grad():
            runner.run_steps(args.steps)
        torch.cuda.synchronize()
        lib.ndcn_prof_enable(0)
        lib.ndcn_prof_read(buf, nk)
        tot_ms = 0.0
        for i, name in enumerate(_lib.PROF_KINDS):
            cnt, ms, byt, fl = buf[4 * i:4 * i + 4]
            if cnt:
                breakdown[name] = {'launches': int(cnt), 'ms_total': round(ms, 3), 'avg_ms': round(ms / cnt, 4),
                                   'GBps': round(byt / ms / 1e6, 1), 'TFLOPs': round(fl / ms / 1e9, 2)}
                tot_ms += ms
        if breakdown:
            dom = max(breakdown, key=lambda k: breakdown[k]['ms_total'])
            i = _lib.PROF_KINDS.index(dom)
            cnt, ms, byt, fl = buf[4 * i:4 * i + 4]
            # which roof is nearer: time the launch would take at the HBM peak vs on the matrix cores.  The fused H = 256
            # kernels issue fp16 MFMAs - 3 products of two-piece splits per fp32 product - so their matrix work is priced as
            # 3 x the Linear's flops at the dense fp16 peak; every other kernel with a GEMM runs the fp32 MFMA.
            t_hbm = byt / cnt / (HBM_PEAK_GBS * 1e9)
            split = dom == 'rhs_fused' and H == 256 and not f.no_control and not f.no_graph
            lin_flops = 2.0 * n_local * H * H if split else 0.0
            if split:
                mfma_peak, mfma_what = MFMA_F16_PEAK_TFLOPS, 'fp16 32x32x16, %d split products per fp32 product' % SPLIT_PRODUCTS
                t_mfma = SPLIT_PRODUCTS * lin_flops / (MFMA_F16_PEAK_TFLOPS * 1e12)
            else:
                mfma_peak, mfma_what = MFMA_F32_PEAK_TFLOPS, 'fp32 32x32x2'
                t_mfma = fl / cnt / (MFMA_F32_PEAK_TFLOPS * 1e12)
            if t_mfma > t_hbm:
                ach = (SPLIT_PRODUCTS * lin_flops * cnt if split else fl) / ms / 1e9
                roofline = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                            'frac': round(ach / mfma_peak, 4)}
            else:
                ach = byt / ms / 1e6
                roofline = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': round(ach / HBM_PEAK_GBS, 4)}
            cfg_name = 'NC' if (args.config == 'M' and args.no_control) else args.config
            if args.method != 'dopri5':
                cfg_name += '_' + args.method                 # its own PMC summary (profiles/*_traffic_pmc_M_euler.json) or none
            traffic, src = pmc_traffic(dom, cfg_name) if (world == 1 and not args.sharded) else (None, None)
            roofline.update({'traffic': traffic, 'traffic_source': src, 'mfma_roof': mfma_what,
                             'traffic_over_algorithmic': round(traffic / (byt / cnt), 3) if traffic else None,
                             'kernel': dom, 'launches': int(cnt),
                             'avg_ms': round(ms / cnt, 4), 'alg_bytes_per_launch': round(byt / cnt),
                             'alg_flops_per_launch': round(fl / cnt),
                             'ms_at_hbm_peak': round(1e3 * t_hbm, 4), 'ms_at_mfma_peak': round(1e3 * t_mfma, 4),
                             'achieved_GBps': round(byt / ms / 1e6, 1), 'achieved_TFLOPs': round(fl / ms / 1e9, 2),
                             'share_of_kernel_time': round(breakdown[dom]['ms_total'] / tot_ms, 3)})

    # device-to-device copy of a 1 GiB panel by the library's own streaming kernel (ndcn_copy_f32: 16 bytes per lane,
    # non-temporal): the HBM rate this box actually sustains (report the fraction of the spec AND of the
    # measured copy; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)
    if roofline is not None:
        from ndcn_amd import hip as _hip
        src_p = torch.empty(256 << 20, dtype=torch.float32, device=dev)
        dst_p = torch.empty_like(src_p)
        _hip.copy(src_p, out=dst_p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _hip.copy(src_p, out=dst_p)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src_p.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        roofline['measured_copy_GBps'] = round(copy_gbps, 1)
        roofline['frac_of_measured_copy'] = round(roofline['achieved_GBps'] / copy_gbps, 4)
        if roofline.get('traffic'):
            # what the HBM actually moved per launch (PMC), over the launch time: how close the kernel runs to the rate
            # a plain device copy sustains on this box
            tg = roofline['traffic'] / (roofline['avg_ms'] * 1e-3) / 1e9
            roofline['traffic_GBps'] = round(tg, 1)
            roofline['traffic_frac_of_measured_copy'] = round(tg / copy_gbps, 4)
        del src_p, dst_p
    halo = None
    if hasattr(runner, 'plan') and (world > 1 or runner.plan.n_halo > 0):
        hb = int(runner.plan.bytes_per_exchange(H))
        # a short instrumented pass: per exchange, its duration on the side stream, the launch it hides behind, and what
        # the main stream still waited for (exposed).  The device-resident solver is timed through its Python twin: the
        # same plan, streams and launches, with events around them.
        timed = runner.python_twin() if hasattr(runner, 'python_twin') else runner
        halo = {'bytes_received_per_rhs_per_gpu': hb, 'exchanges_per_step': 6,
                'xgmi_peak_GBps_per_gpu': 7 * 153, 'overlapped_with_interior_rows': bool(timed.func.overlap),
                'two_phase_own_columns_under_exchange': bool(timed.func.two_phase),
                'solver': 'device-resident (C ABI, library-owned RCCL communicator)' if timed is not runner else 'python-stepped over torch.distributed'}
        timed.func.timing = {}
        with torch.no_grad():
            timed.run_steps(min(args.steps, 5))
        torch.cuda.synchronize()
        tm = timed.func.drain_timing() or {}
        timed.func.timing = None
        if tm.get('n'):
            halo.update({'exchange_us': round(tm['exchange_us'] / tm['n'], 1), 'interior_us': round(tm['interior_us'] / tm['n'], 1),
                         'exposed_us': round(tm['exposed_us'] / tm['n'], 1), 'timed_exchanges': tm['n']})

    if rank != 0:
        return
    out = {
        'metric': 'node-states/sec (N x T_steps), 1M-node grid H=256' if (args.config == 'M' and args.method == 'dopri5')
                  else 'node-states/sec (N x T_steps), config %s (parity / measurement case, not the judged line)'
                       % (args.config if args.config != 'M' else 'M with --method ' + args.method),
        'value': round(value, 1),
        'unit': 'node-states/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(1e3 * wall / args.steps, 4),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,                                   # the reference publishes no number for this metric
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': what,
                   'parallelism': 'single GPU' if world == 1 else 'node-range sharding x%d + RCCL halo exchange per RHS' % world,
                   'step': step_desc},
        'node_rhs_per_s': round(n_total * nfe / wall, 1),
        'rhs_evals': nfe,
        'roofline': roofline,
        'kernels': breakdown,
        'spmm_standalone': spmm_line,
        'device': device_info(),
        'halo_exchange': halo,
    }
    if world == 1 and not args.sharded and not args.no_cpu_baseline:
        cfg_name = 'NC' if (args.config == 'M' and args.no_control) else args.config
        out['cpu_baseline'], out['parity'] = cpu_baseline(cfg_name, H, args.T, args.rtol, args.atol, args.cpu_threads,
                                                           args.cpu_runs, args.cpu_nodes, dev=dev, at_scale=not args.no_at_scale)
    else:
        out['cpu_baseline'] = out['parity'] = None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
