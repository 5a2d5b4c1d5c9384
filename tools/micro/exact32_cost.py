#!/usr/bin/env python3
"""What the range guard's fp32 route costs (csrc/rhs.hip: NDCN_PATH_EXACT32): the device-resident dopri5 solver on the metric's lattice
(default 1000 x 1000, H = 256) with ordinary weights (split-fp16 product, rhs_fused3) and with the same weights after one row has been
spread over 2^24 (the guard sends every launch to rhs_fused_256_kernel + composed stage kernels).  ms per attempted step, median of
`reps` solves of 4 attempts; the path bits of the last right-hand side.

    python tools/micro/exact32_cost.py [side] [reps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from ndcn_amd import graphs, _lib
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    dev = torch.device('cuda:0')
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = graphs.to_device(L, dev)
    n, H = side * side, 256
    torch.manual_seed(0)
    f = ODEFunc(H, A).to(dev).eval()
    x0 = torch.rand(n, H, device=dev)
    out = torch.empty(1, n, H, device=dev)
    for label in ('ordinary weights', 'one weight row spread over 2^24'):
        if label != 'ordinary weights':
            with torch.no_grad():
                f.wt.weight[17, :] *= 2.0 ** -20
                f.wt.weight[17, 33] = 0.05
        times, steps = [], 0
        for r in range(reps + 1):
            solver = DeviceSolver(f, n, 'dopri5', .01, .001)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                solver.begin(x0, 0.0)
                solver.advance_many([5.0], out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            steps = len(solver.steplog())
            if r:
                times.append(1e3 * dt / steps)
            solver.close()
        print(json.dumps({'case': '%dx%d lattice, H=256, dopri5 rtol .01 atol .001 to t=5 (whole solve incl. initial step and dense output)' % (side, side),
                          'weights': label, 'attempted_steps': steps, 'ms_per_step_whole_solve': round(float(np.median(times)), 3),
                          'rhs_path_bits': int(_lib.load().ndcn_debug_last_rhs_path())}), flush=True)


if __name__ == '__main__':
    main()
