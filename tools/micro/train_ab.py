"""A/B of the dopri5 training path's switches inside ONE box (host-bound cases differ by 50 % between boxes):
   python tools/micro/train_ab.py  -> ms per Adam step of the README-sized dopri5 case and the 100k-node case per variant"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = """
import sys, json, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools')
import bench_train
dev = torch.device('cuda:0')
out = {}
out['c1'] = bench_train.one_case('c1', 20, 20, 80, 'dopri5', dev, 0)['gpu_ms_per_adam_step']
out['100k'] = bench_train.one_case('100k', 316, 256, 10, 'dopri5', dev, 0)['gpu_ms_per_adam_step']
print('RESULT', json.dumps(out))
""" % (ROOT, ROOT)
VARIANTS = [('default', {}), ('separate error', {'NDCN_GRAD_FUSED_ERROR': '0'}), ('eager scalars', {'NDCN_GRAD_LAZY': '0'}), ('unfused stages', {'NDCN_GRAD_FUSED_STAGE': '0'}),
            ('both off', {'NDCN_GRAD_LAZY': '0', 'NDCN_GRAD_FUSED_STAGE': '0'}), ('default again', {})]
for name, env in VARIANTS:
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', CODE], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
    print('%-16s %s' % (name, line[0][7:] if line else r.stderr[-300:]), flush=True)
