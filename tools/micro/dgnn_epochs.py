"""N epochs of the README's dgnn command on Cora (for rocprofv3 kernel statistics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_dgnn as b
dev = torch.device('cuda:0')
case = b.load_case('cora')
model, opt, x, y, itr, iva = b.build_hip(case, dev)
for _ in range(int(os.environ.get('EPOCHS', 40))):
    b.epoch_hip(model, opt, x, y, itr, iva)
torch.cuda.synchronize()
