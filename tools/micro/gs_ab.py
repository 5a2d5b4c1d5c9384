"""A/B of the gS = gZ W kernels (NDCN_GS_ROWS = 0: resident weights, 32 / 64: tile kernels): time at n = 10^5 and a checksum of the
result on fixed inputs (ragged n, outlier operands) - the three forms must print the same checksums."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ndcn_amd import hip
dev = torch.device('cuda:0')
H = 256
for n in (99856, 100003, 37, 1):
    gen = torch.Generator().manual_seed(n)
    g = torch.randn(n, H, generator=gen); g[:, 5] *= 4096.0
    W = torch.randn(H, H, generator=gen) / 16; W[17, 33] *= 4096.0
    Y = torch.rand(n, H, generator=gen) - 0.3
    gS, _, _ = hip.linear_bwd(g.to(dev), W.to(dev), S=None, Y=Y.to(dev), need_gS=True, need_gW=False, need_gb=False)
    print('n', n, 'sha', hashlib.sha256(gS.cpu().numpy().tobytes()).hexdigest()[:16])
n = 99856
g = torch.randn(n, H, device=dev); W = torch.randn(H, H, device=dev) / 16; Y = torch.relu(torch.randn(n, H, device=dev))
for _ in range(3): hip.linear_bwd(g, W, S=None, Y=Y, need_gW=False, need_gb=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): hip.linear_bwd(g, W, S=None, Y=Y, need_gW=False, need_gb=False)
b.record(); torch.cuda.synchronize()
print('gS %.4f ms' % (a.elapsed_time(b) / 50))
