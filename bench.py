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

Prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed on the
launch stream during a second, instrumented pass over the same K steps) and `cpu_baseline` (the CPU oracle
timed on a bounded sample, rank 0, N = 1 only).
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


KERNEL_FAMILY = {'rhs_fused': 'rhs_fused', 'spmm': 'spmm_', 'combine': 'combine_kernel', 'linear': 'linear_',
                 'error': 'rk_error_kernel'}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary (profiles/*traffic_pmc.json,
    produced by tools/gpu_final.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Launch-weighted mean over the family's kernels; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*traffic_pmc.json')))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))['kernels']
    except Exception:
        return None, None
    num = den = 0
    for name, v in d.items():
        if family in name and 'finish' not in name:
            num += v['hbm_bytes_per_launch'] * v['launches']
            den += v['launches']
    return (int(num / den), os.path.relpath(files[-1], ROOT)) if den else (None, None)


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
    p.add_argument('--cpu-side', type=int, default=512, help='grid side of the bounded CPU-baseline sample')
    p.add_argument('--no-profile-pass', action='store_true')
    p.add_argument('--sharded', action='store_true', help='force the multi-GPU code path (needs torchrun, works with 1 rank)')
    return p.parse_args()


class SingleGpuRunner:
    """Counts attempted dopri5 steps of the device-resident solver, restarting at t = T."""

    def __init__(self, f, x0, T, rtol, atol):
        from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
        self.solver = DeviceSolver(f, x0.shape[0], 'dopri5', rtol, atol)
        self.x0, self.T = x0, T
        self.out = torch.empty_like(x0)
        self.solver.begin(x0, 0.0)
        self.restarts = 0
        self.nfe_done = 0

    def run_steps(self, k):
        done = 0
        while done < k:
            before = self.solver.stats()['steps']
            reached = self.solver.advance(self.T, self.out, step_budget=k - done)
            done += int(self.solver.stats()['steps'] - before)
            if reached:
                self.nfe_done += int(self.solver.stats()['nfe'])
                self.solver.begin(self.x0, 0.0)          # resets the solver's own counters
                self.restarts += 1
        return done

    def nfe(self):
        return self.nfe_done + int(self.solver.stats()['nfe'])


def cpu_baseline(side, H, T, rtol, atol, threads):
    """The CPU oracle (torch-CPU restatement of the reference path: torch.sparse.mm on COO + F.linear + the
    restated dopri5 loop) on a bounded sample of the same workload: one solve on a side x side grid."""
    from ndcn_amd import graphs
    from oracle import ndcn_oracle as orc
    # the reference's op-per-term solver issues ~300 small tensor ops per step: beyond a few dozen threads the
    # fork/join cost of each op outweighs the work, so the leg uses a bounded thread count and says which
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
    torch.manual_seed(0)
    lin = torch.nn.Linear(H, H)
    f = orc.OracleODEFunc(A, lin.weight.detach(), lin.bias.detach())
    x0 = torch.rand(side * side, H)
    log = []
    t0 = time.perf_counter()
    orc.odeint(f, x0, torch.tensor([0., T]), rtol=rtol, atol=atol, method='dopri5', step_log=log)
    dt = time.perf_counter() - t0
    n = side * side
    return {'value': n * len(log) / dt, 'unit': 'node-states/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'one dopri5 solve t in [0,%g] on a %dx%d grid (N=%d, H=%d): %d steps, %d RHS evals, %.1f s'
                      % (T, side, side, n, H, len(log), f.nfe, dt)}


def main():
    args = parse()
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
    n_local = S * S
    torch.manual_seed(0)
    f = ODEFunc(H, None).to(dev).eval()                      # nn.Linear default init, seed 0
    if world == 1 and not args.sharded:
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(S))
        f.A = graphs.to_device(L, dev)
        nnz = int(L.nnz)
        x0 = torch.rand(n_local, H, generator=torch.Generator().manual_seed(0)).to(dev)
        runner = SingleGpuRunner(f, x0, args.T, args.rtol, args.atol)
    else:
        from ndcn_amd.sharding import ShardedGridBench
        runner = ShardedGridBench(f, S, world, rank, dev, args.T, args.rtol, args.atol)
        nnz = runner.local_nnz

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
            # which roof is nearer: time the launch would take at the HBM peak vs at the fp32 MFMA peak
            t_hbm = byt / cnt / (HBM_PEAK_GBS * 1e9)
            t_mfma = fl / cnt / (MFMA_F32_PEAK_TFLOPS * 1e12)
            if t_mfma > t_hbm:
                ach = fl / ms / 1e9
                roofline = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4)}
            else:
                ach = byt / ms / 1e6
                roofline = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': round(ach / HBM_PEAK_GBS, 4)}
            traffic, src = pmc_traffic(KERNEL_FAMILY.get(dom, dom))
            roofline.update({'traffic': traffic, 'traffic_source': src, 'kernel': dom, 'launches': int(cnt),
                             'avg_ms': round(ms / cnt, 4), 'alg_bytes_per_launch': round(byt / cnt),
                             'alg_flops_per_launch': round(fl / cnt),
                             'ms_at_hbm_peak': round(1e3 * t_hbm, 4), 'ms_at_mfma_peak': round(1e3 * t_mfma, 4),
                             'achieved_GBps': round(byt / ms / 1e6, 1), 'achieved_TFLOPs': round(fl / ms / 1e9, 2),
                             'share_of_kernel_time': round(breakdown[dom]['ms_total'] / tot_ms, 3)})

    # device-to-device copy of one panel: the HBM rate this box actually sustains (report the fraction of
    # the spec AND of the measured copy)
    if roofline is not None:
        src_p = torch.empty(256 << 20, dtype=torch.float32, device=dev)
        dst_p = torch.empty_like(src_p)
        dst_p.copy_(src_p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst_p.copy_(src_p)
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
    if world > 1 and hasattr(runner, 'plan'):
        hb = int(runner.plan.bytes_per_exchange(H))
        halo = {'bytes_received_per_rhs_per_gpu': hb, 'exchanges_per_step': 6,
                'xgmi_peak_GBps_per_gpu': 7 * 153}

    if rank != 0:
        return
    out = {
        'metric': 'node-states/sec (N x T_steps), 1M-node grid H=256',
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
        'config': {'workload': 'NDCN ODEFunc relu(W(AX)+b), %dx%d 8-neighbour grid per GPU (N=%d nodes total), '
                               'normalised-Laplacian CSR nnz=%d per GPU, H=%d, dopri5 rtol=%g atol=%g t in [0,%g], '
                               'state X~U(0,1) seed 0, nn.Linear default init seed 0'
                               % (S, S, n_total, nnz, H, args.rtol, args.atol, args.T),
                   'parallelism': 'single GPU' if world == 1 else 'node-range sharding x%d + RCCL halo exchange per RHS' % world,
                   'step': 'one attempted dopri5 step (6 RHS evals + stage algebra + error norm + controller)'},
        'node_rhs_per_s': round(n_total * nfe / wall, 1),
        'rhs_evals': nfe,
        'roofline': roofline,
        'kernels': breakdown,
        'device': device_info(),
        'halo_exchange': halo,
    }
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args.cpu_side, H, args.T, args.rtol, args.atol, args.cpu_threads)
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
