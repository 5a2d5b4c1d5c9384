"""Host-side profile of the sharded (Python-driven) solver loop: where the CPU time between kernels goes."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29721')
os.environ.setdefault('NDCN_SELF_HALO', '2000')
import torch
import torch.distributed as dist
dev = torch.device('cuda:0')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from ndcn_amd.neural_dynamics import ODEFunc
from ndcn_amd.sharding import ShardedGridBench
torch.manual_seed(0)
f = ODEFunc(256, None).to(dev).eval()
with torch.no_grad():
    r = ShardedGridBench(f, 1000, 1, 0, dev, 5.0, .01, .001)
    r.run_steps(6)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    r.run_steps(12)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr).sort_stats('cumulative')
st.print_stats(28)
dist.destroy_process_group()
