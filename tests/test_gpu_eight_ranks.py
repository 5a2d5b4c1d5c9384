"""BASELINE config 4 at its stated cut count, and the metric's grid at 8 shards, on the ONE device of the test box.

Eight processes on cuda:0 (gloo with host staging: RCCL refuses two ranks per device) run the N > 1 product path -
HaloPlan at 8 cuts with 7 peers, the halo exchange, the two-phase form ([halo | hub] layout where hubs exist) or the
row-split form, the HALO variants of the fused kernels, the global error reduction - on the generators of
gene_dynamics.py:99-103 (Newman-Watts-Strogatz k = 5, p = 0.5) and utils_in_learn_dynamics.py:137-157 (grid).  The stitched
rk4 and dopri5 trajectories and the accept / reject log must equal the unsharded device-resident solver on the same
graph (which the other GPU tests pin to the oracle).

Sizes: NDCN_C4_TEST_NODES nodes per rank (default 500 000 = the configuration's own 4M
nodes; the run's record is committed under profiles/), grid 8 x (500 x 1000) = 4M nodes.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

H = 256
WORLD = 8
RTOL, ATOL = 1e-2, 1e-3


def _ticks(case):
    # short spans: every evaluation moves ~3 GB of halo rows through gloo's host staging in the small-world case
    return [0., 0.25] if case == 'small_world' else [0., 0.5]


def _graph(case, per_rank):
    from ndcn_amd import graphs
    if case == 'small_world':
        n = per_rank * WORLD
        L = graphs.normalized_laplacian(graphs.make_graph('small_world', n, seed=0)).tocsr()
        return L, [(n * r) // WORLD for r in range(WORLD + 1)]
    R, C = per_rank // 1000, 1000                                # `R` lattice rows of 1000 nodes per rank
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor_rect(R * WORLD, C)).tocsr()
    return L, [r * R * C for r in range(WORLD + 1)]


def _x_block(rank, n_local):
    return torch.rand(n_local, H, generator=torch.Generator().manual_seed(100 + rank))


def _worker(rank, world, port, case, shm, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('OMP_NUM_THREADS', '4')
    import scipy.sparse as sp
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ndcn_amd import sharding, hip, _lib
        from ndcn_amd.neural_dynamics import ODEFunc
        dev = torch.device('cuda:0')
        bounds = np.load(os.path.join(shm, 'bounds.npy')).tolist()
        lo, hi = bounds[rank], bounds[rank + 1]
        indptr = np.load(os.path.join(shm, 'indptr.npy'), mmap_mode='r')
        a, b = int(indptr[lo]), int(indptr[hi])
        block = sp.csr_matrix((np.load(os.path.join(shm, 'data.npy'), mmap_mode='r')[a:b],
                               np.load(os.path.join(shm, 'indices.npy'), mmap_mode='r')[a:b],
                               np.asarray(indptr[lo:hi + 1]) - a), shape=(hi - lo, bounds[-1]))
        f = ODEFunc(H, None).to(dev)
        f.load_state_dict({'wt.weight': torch.from_numpy(np.load(os.path.join(shm, 'W.npy'))),
                           'wt.bias': torch.from_numpy(np.load(os.path.join(shm, 'b.npy')))})
        plan = sharding.HaloPlan(block, bounds, rank, dev)
        xl = _x_block(rank, hi - lo).to(dev)
        t = torch.tensor(_ticks(case), device=dev)
        out = {'n_halo': plan.n_halo, 'peers_sending': int(sum(1 for c in plan.recv_counts if c > 0)),
               'ranges': None if plan.ranges is None else [(int(r[0]), int(r[1]), bool(r[3])) for r in plan.ranges]}
        with torch.no_grad():
            for method in ('rk4', 'dopri5'):
                log, st = [], {}
                y = sharding.sharded_odeint(hip, f, plan, bounds[-1], xl, t, rtol=RTOL, atol=ATOL, method=method,
                                            step_log=log, stats=st)
                np.save(os.path.join(shm, 'y_%s_%d.npy' % (method, rank)), y[-1].cpu().numpy())
                out[method] = {'form': st['form'], 'nfe': st['nfe'], 'halo_bytes_received_per_rhs': 4 * H * plan.n_halo,
                               'log': [tuple(float(v) for v in r[:4]) for r in log if r[0] != 'nfe']}
                del y
        out['path'] = int(_lib.load().ndcn_debug_last_rhs_path())
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['small_world', 'grid'])
def test_eight_shards_on_one_device_equal_the_unsharded_solver(case):
    import torch.multiprocessing as mp
    from ndcn_amd import CsrOperator, _lib
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq import odeint
    dev = torch.device('cuda:0')
    per_rank = int(os.environ.get('NDCN_C4_TEST_NODES', '500000')) if case == 'small_world' else 500000
    L, bounds = _graph(case, per_rank)
    n = L.shape[0]
    torch.manual_seed(0)
    f = ODEFunc(H, None)
    shm = tempfile.mkdtemp(prefix='ndcn8_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        for name, arr in (('indptr', L.indptr), ('indices', L.indices), ('data', L.data), ('bounds', np.asarray(bounds)),
                          ('W', f.wt.weight.detach().numpy()), ('b', f.wt.bias.detach().numpy())):
            np.save(os.path.join(shm, name + '.npy'), arr)
        # (a SPAWNED manager: a fork()ed server inherits this process's garbage - tensors, events, handles of a HIP context it does
        # not have - and its garbage collector then runs their destructors)
        ret = mp.get_context('spawn').Manager().dict()
        port = 29300 + os.getpid() % 300 + (0 if case == 'grid' else 301)
        mp.spawn(_worker, args=(WORLD, port, case, shm, ret), nprocs=WORLD, join=True)
        assert len(ret) == WORLD
        # the evaluation form on EVERY rank, and what travelled
        for r in range(WORLD):
            o = ret[r]
            assert o['n_halo'] > 0
            if case == 'small_world':
                assert o['ranges'] is None and o['peers_sending'] == WORLD - 1          # shortcuts point anywhere: 7 peers
                assert o['rk4']['form'] == o['dopri5']['form'] == 'two_phase'
                assert o['path'] & _lib.PATH_HALO
            else:
                assert o['rk4']['form'] == o['dopri5']['form'] == 'row_split'
                kinds = [k for _, _, k in o['ranges']]
                assert kinds == ([False, True] if r == 0 else [True, False] if r == WORLD - 1 else [True, False, True])
                assert o['n_halo'] == (1000 if r in (0, WORLD - 1) else 2000)          # one lattice row per cut
                assert o['path'] & _lib.PATH_FUSED3                                    # the blocks keep the lattice plan
            assert o['dopri5']['log'] == ret[0]['dopri5']['log']                       # identical decisions on all ranks
        # the unsharded device-resident solver on the same graph, after the workers have left the device
        A = CsrOperator.from_arrays(L.indptr, L.indices, L.data, L.shape, dev)
        fd = ODEFunc(H, A).to(dev).eval()
        fd.load_state_dict(f.state_dict())
        x = torch.cat([_x_block(r, bounds[r + 1] - bounds[r]) for r in range(WORLD)]).to(dev)
        t = torch.tensor(_ticks(case), device=dev)
        record = {'case': case, 'world': WORLD, 'nodes': n, 'nnz': int(L.nnz), 'H': H, 'ticks': _ticks(case),
                  'rtol': RTOL, 'atol': ATOL, 'ranks': {}}
        with torch.no_grad():
            for method in ('rk4', 'dopri5'):
                log = []
                ref = odeint(fd, x, t, rtol=RTOL, atol=ATOL, method=method, step_log=log)[-1]
                worst = 0.0
                for r in range(WORLD):
                    got = torch.from_numpy(np.load(os.path.join(shm, 'y_%s_%d.npy' % (method, r)))).to(dev)
                    worst = max(worst, float((got - ref[bounds[r]:bounds[r + 1]]).abs().max()))
                    del got
                scale = max(1.0, float(ref.abs().max()))
                record[method + '_max_abs_diff'] = worst
                assert worst < 2e-5 * scale, (method, worst)
                if method == 'dopri5':
                    rows = [r for r in log if r[0] != 'nfe']
                    mine = ret[0]['dopri5']['log']
                    assert len(rows) == len(mine) >= 1
                    assert [bool(r[2]) for r in rows] == [bool(m[2]) for m in mine]           # accept / reject sequence
                    assert np.allclose([r[1] for r in rows], [m[1] for m in mine], rtol=1e-5)   # step sizes
                    record['dopri5_attempts'], record['dopri5_accepts'] = len(rows), int(sum(bool(r[2]) for r in rows))
                del ref
        for r in range(WORLD):
            o = ret[r]
            record['ranks'][r] = {'form': o['dopri5']['form'], 'halo_rows': o['n_halo'], 'peers': o['peers_sending'],
                                  'halo_bytes_received_per_rhs': o['dopri5']['halo_bytes_received_per_rhs'],
                                  'rhs_evals': o['dopri5']['nfe'] + o['rk4']['nfe'], 'path_bits': o['path']}
        out_dir = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, 'eight_ranks_%s_%d.json' % (case, per_rank)), 'w') as fh:
                json.dump(record, fh, indent=1)
    finally:
        shutil.rmtree(shm, ignore_errors=True)


# ---- the DEVICE-RESIDENT sharded solver (ndcn_solver_desc::shard: exchange, launches, all-reduce, controller inside the library) with
# world = 8 on the one device, over the loopback transport (include/ndcn_hip.h: ndcn_comm_create_loopback - the RCCL communicator's
# call sequence on shared memory; RCCL itself refuses two ranks per device)

def _device_worker(rank, world, port, case, shm, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('OMP_NUM_THREADS', '4')
    import scipy.sparse as sp
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ndcn_amd import sharding, hip
        from ndcn_amd.neural_dynamics import ODEFunc
        from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
        dev = torch.device('cuda:0')
        bounds = np.load(os.path.join(shm, 'bounds.npy')).tolist()
        lo, hi = bounds[rank], bounds[rank + 1]
        indptr = np.load(os.path.join(shm, 'indptr.npy'), mmap_mode='r')
        a, b = int(indptr[lo]), int(indptr[hi])
        block = sp.csr_matrix((np.load(os.path.join(shm, 'data.npy'), mmap_mode='r')[a:b],
                               np.load(os.path.join(shm, 'indices.npy'), mmap_mode='r')[a:b],
                               np.asarray(indptr[lo:hi + 1]) - a), shape=(hi - lo, bounds[-1]))
        f = ODEFunc(H, None).to(dev)
        f.load_state_dict({'wt.weight': torch.from_numpy(np.load(os.path.join(shm, 'W.npy'))),
                           'wt.bias': torch.from_numpy(np.load(os.path.join(shm, 'b.npy')))})
        plan = sharding.HaloPlan(block, bounds, rank, dev)
        xl = _x_block(rank, hi - lo).to(dev)
        ticks = [0., 0.3, 0.6]
        shard = sharding.DeviceShard(plan, bounds[-1], transport='loopback')
        out = {'transport': shard.transport, 'n_halo': plan.n_halo}
        with torch.no_grad():
            solver = DeviceSolver(f, hi - lo, 'dopri5', RTOL, ATOL, shard=shard)
            yd = torch.empty(2, hi - lo, H, device=dev)
            solver.begin(xl, 0.0)
            solver.advance_many(ticks[1:], yd)
            torch.cuda.synchronize()
            out['dopri5_log'] = [tuple(float(v) for v in r[:4]) for r in solver.steplog()]
            out['dopri5_nfe'] = int(solver.stats()['nfe'])
            np.save(os.path.join(shm, 'yd_dopri5_%d.npy' % rank), yd[-1].cpu().numpy())
            rk = DeviceSolver(f, hi - lo, 'rk4', shard=shard)
            y4 = torch.empty(2, hi - lo, H, device=dev)
            rk.begin(xl, 0.0)
            rk.advance_many(ticks[1:], y4)
            torch.cuda.synchronize()
            np.save(os.path.join(shm, 'yd_rk4_%d.npy' % rank), y4[-1].cpu().numpy())
            # the Python-stepped form of the same shard (gloo): the two implementations of the N > 1 path against each other
            lp = []
            yp = sharding.sharded_odeint(hip, f, plan, bounds[-1], xl, torch.tensor(ticks, device=dev), rtol=RTOL, atol=ATOL,
                                         method='dopri5', step_log=lp)
            out['python_log'] = [tuple(float(v) for v in r[:4]) for r in lp if r[0] != 'nfe']
            out['device_vs_python_max_abs'] = float((yd[-1] - yp[-1]).abs().max())
            solver.close(); rk.close()
        shard.close()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['small_world', 'grid'])
def test_eight_shards_device_resident_solver_over_the_loopback_transport(case):
    import torch.multiprocessing as mp
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq import odeint
    dev = torch.device('cuda:0')
    per_rank = 40000 if case == 'small_world' else 64000         # (host-staged: every byte crosses PCIe twice and shared memory once)
    L, bounds = _graph(case, per_rank)
    torch.manual_seed(0)
    f = ODEFunc(H, None)
    shm = tempfile.mkdtemp(prefix='ndcn8d_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        for name, arr in (('indptr', L.indptr), ('indices', L.indices), ('data', L.data), ('bounds', np.asarray(bounds)),
                          ('W', f.wt.weight.detach().numpy()), ('b', f.wt.bias.detach().numpy())):
            np.save(os.path.join(shm, name + '.npy'), arr)
        ret = mp.get_context('spawn').Manager().dict()
        port = 29000 + os.getpid() % 250 + (0 if case == 'grid' else 251)
        mp.spawn(_device_worker, args=(WORLD, port, case, shm, ret), nprocs=WORLD, join=True)
        assert len(ret) == WORLD
        for r in range(WORLD):
            o = ret[r]
            assert o['transport'] == 'loopback' and o['n_halo'] > 0
            assert o['dopri5_log'] == ret[0]['dopri5_log'] and len(o['dopri5_log']) >= 1          # identical decisions on all ranks
            assert [bool(x[2]) for x in o['dopri5_log']] == [bool(x[2]) for x in o['python_log']]    # ... and in both implementations
            assert o['device_vs_python_max_abs'] < 1e-5
        A = CsrOperator.from_arrays(L.indptr, L.indices, L.data, L.shape, dev)
        fd = ODEFunc(H, A).to(dev).eval()
        fd.load_state_dict(f.state_dict())
        x = torch.cat([_x_block(r, bounds[r + 1] - bounds[r]) for r in range(WORLD)]).to(dev)
        t = torch.tensor([0., 0.3, 0.6], device=dev)
        with torch.no_grad():
            for method in ('rk4', 'dopri5'):
                log = []
                ref = odeint(fd, x, t, rtol=RTOL, atol=ATOL, method=method, step_log=log)[-1]
                scale = max(1.0, float(ref.abs().max()))
                for r in range(WORLD):
                    got = torch.from_numpy(np.load(os.path.join(shm, 'yd_%s_%d.npy' % (method, r)))).to(dev)
                    assert float((got - ref[bounds[r]:bounds[r + 1]]).abs().max()) < 2e-5 * scale, (method, r)
                if method == 'dopri5':
                    rows = [r for r in log if r[0] != 'nfe']
                    assert [bool(r[2]) for r in rows] == [bool(m[2]) for m in ret[0]['dopri5_log']]
    finally:
        shutil.rmtree(shm, ignore_errors=True)
