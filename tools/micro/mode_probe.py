"""Probe aid for variant builds of rhs_fused3.hip: every epilogue mode of the fused RHS in a process of its own (40 s limit each),\nso that a faulting or hanging variant names itself:  python tools/micro/mode_probe.py [grid side]"""
import sys, subprocess, os
MODES = ['plain', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'c4aux', 'e1', 'e5', 'rk4_0', 'rk4_3']
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    mode, side = sys.argv[2], int(sys.argv[3])
    import torch, numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from ndcn_amd import hip, graphs, _lib
    dev = torch.device('cuda:0')
    A = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(side)), dev)
    n = side * side; H = 256
    g = torch.Generator().manual_seed(0)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W = ((torch.rand(H, H, generator=g) - .5) / 8).to(dev); b = ((torch.rand(H, generator=g) - .5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    ref = hip.rhs(A, X, W, b) if mode != 'plain' else None
    for rep in range(3):
        if mode == 'plain': K = hip.rhs(A, X, W, b)
        elif mode.startswith('c') and mode.endswith('aux'):
            K, yn, E = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:4], cs[:4] + [cs[5]], aux_cs=cs[:4] + [cs[4]])
        elif mode.startswith('c'):
            m = int(mode[1:]); K, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:m], cs[:m] + [cs[5]])
        elif mode.startswith('e'):
            m = int(mode[1:]); K, (s_, b_) = hip.rhs_rk(A, X, W, b, 'error', y0, ks[:m], cs[:m] + [cs[5]], rtol=1e-2, atol=1e-3)
        else:
            m = int(mode.split('_')[1]); K, yn = hip.rhs_rk(A, X, W, b, 'rk4', y0, ks[:m], [np.float32(0.1)])
        torch.cuda.synchronize()
    print('OK', mode, 'path', _lib.load().ndcn_debug_last_rhs_path(), float(K.abs().sum()))
else:
    side = sys.argv[1] if len(sys.argv) > 1 else '200'
    for m in MODES:
        try:
            r = subprocess.run([sys.executable, __file__, 'one', m, side], capture_output=True, text=True, timeout=40)
            print(m, 'rc', r.returncode, (r.stdout.strip().splitlines() or ['-'])[-1][:80], flush=True)
        except subprocess.TimeoutExpired:
            print(m, 'TIMEOUT', flush=True)
