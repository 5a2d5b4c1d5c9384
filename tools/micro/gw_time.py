"""Timing aid: ndcn_linear_bwd_f32 at n = 10^5, H = 256 (gS / gW+gb separately), HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ndcn_amd import hip
dev = torch.device('cuda:0')
n, H = 99856, 256
g = torch.randn(n, H, device=dev); S = torch.randn(n, H, device=dev); W = torch.randn(H, H, device=dev) / 16
Y = torch.relu(torch.randn(n, H, device=dev))
for what, kw in (('gS', dict(need_gW=False, need_gb=False)), ('gW+gb', dict(need_gS=False))):
    for _ in range(3): hip.linear_bwd(g, W, S=S, Y=Y, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hip.linear_bwd(g, W, S=S, Y=Y, **kw)
    b.record(); torch.cuda.synchronize()
    print(what, '%.3f ms' % (a.elapsed_time(b) / 20))
