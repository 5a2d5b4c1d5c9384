#!/usr/bin/env python3
"""In which order does ATen's float32 `Tensor.norm()` (torch CPU) add up?  The reference's dopri5 initial step divides
three such norms (misc.py:71-76,121-138), and at rtol 1e-7 a 1-ulp difference of the first step size reshuffles later
accept / reject decisions.  This script matches torch's result bit for bit against emulated summation orders on random
vectors; on torch 2.10 (x86, AVX-512 capable host) exactly one candidate matches 60 / 60:

    8 running sums (lane j owns elements j, j+8, ...), acc_j = fma(x, x, acc_j) in index order; the eight sums added
    left to right; the n % 8 tail elements added with fma; sqrt.          (thread count does not matter: the path is serial)

ndcn_amd/csrc/rk.hip: scaled_sumsq_aten_kernel forms the initial-step norms in that order for panels <= 2^20 elements."""
import numpy as np
import torch

f32 = np.float32


def lanes(x, W, fma):
    m = len(x) - (len(x) % W)
    acc = np.zeros(W, dtype=np.float32)
    for b in x[:m].reshape(-1, W):
        acc = (acc.astype(np.float64) + b.astype(np.float64) * b.astype(np.float64)).astype(np.float32) if fma else acc + b * b
    return acc, x[m:]


def hsum(a, how):
    a = a.copy()
    if how == 'seq':
        s = a[0]
        for v in a[1:]:
            s = f32(s + v)
        return s
    while len(a) > 1:
        a = (a[:len(a) // 2] + a[len(a) // 2:]).astype(np.float32) if how == 'tree' else (a[0::2] + a[1::2]).astype(np.float32)
    return a[0]


def main():
    torch.set_num_threads(1)
    rng = np.random.RandomState(1)
    cases = []
    for _ in range(60):
        n = int(rng.choice([8000, 7, 65, 12345, 300001, 54321, 1023, 100]))
        x = (rng.randn(n) * rng.choice([1e-3, 1, 50])).astype(np.float32)
        cases.append((x, torch.from_numpy(x).norm().item()))
    for W in (8, 16, 32):
        for fma in (0, 1):
            for how in ('seq', 'tree', 'adjacent'):
                for tail_fma in (0, 1):
                    ok = 0
                    for x, t in cases:
                        acc, tl = lanes(x, W, fma)
                        s = hsum(acc, how)
                        for v in tl:
                            s = f32(np.float64(s) + np.float64(v) * np.float64(v)) if tail_fma else f32(s + f32(v * v))
                        ok += float(f32(np.sqrt(s))) == t
                    print('lanes %2d  fma %d  lane-sum %-8s tail-fma %d : %2d / %d' % (W, fma, how, tail_fma, ok, len(cases)))


if __name__ == '__main__':
    main()
