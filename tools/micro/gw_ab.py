"""A/B of the gW = gZ^T S kernels (NDCN_GW_WS = 1: role-split, 0: the phase-alternating kernel): time at n = 10^5 and checksums of gW /
gb on fixed inputs (ragged n, outlier operands) - gW must print the same checksum under both settings."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ndcn_amd import hip
dev = torch.device('cuda:0')
H = 256
sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
for n in (99856, 100003, 37, 1, 5000):
    gen = torch.Generator().manual_seed(n)
    g = torch.randn(n, H, generator=gen); g[:, 5] *= 4096.0
    S = torch.randn(n, H, generator=gen); S[:, 9] *= 1e-6
    W = torch.randn(H, H, generator=gen) / 16
    Y = torch.rand(n, H, generator=gen) - 0.3
    _, gW, gb = hip.linear_bwd(g.to(dev), W.to(dev), S=S.to(dev), Y=Y.to(dev), need_gS=False)
    ref = (g * (Y > 0)).double().t() @ S.double()
    print('n', n, 'gW', sha(gW), 'gb', sha(gb), 'gW vs fp64 %.2e' % float((gW.cpu().double() - ref).abs().max() / ref.abs().max()),
          'gb vs fp64 %.2e' % float((gb.cpu().double() - (g * (Y > 0)).double().sum(0)).abs().max()))
n = 99856
g = torch.randn(n, H, device=dev); S = torch.randn(n, H, device=dev); W = torch.randn(H, H, device=dev) / 16; Y = torch.relu(torch.randn(n, H, device=dev))
for _ in range(3): hip.linear_bwd(g, W, S=S, Y=Y, need_gS=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): hip.linear_bwd(g, W, S=S, Y=Y, need_gS=False)
b.record(); torch.cuda.synchronize()
print('gW+gb %.4f ms' % (a.elapsed_time(b) / 50))
