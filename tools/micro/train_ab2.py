"""A/B inside ONE box (boxes differ by 5-10 %): 100k-node dopri5 Adam step under environment variants given as arguments 'K=V,K=V'."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = """
import sys, json, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools')
import bench_train
r = bench_train.one_case('100k', 316, 256, 10, 'dopri5', torch.device('cuda:0'), 0)
print('RESULT', r['gpu_ms_per_adam_step'], r.get('library_kernel_ms_per_step'), {k: v['ms_total'] for k, v in r['breakdown'].items() if v['ms_total'] > 1})
""" % (ROOT, ROOT)
for spec in sys.argv[1:]:
    e = dict(os.environ)
    if spec != '-':
        e.update(dict(kv.split('=') for kv in spec.split(',')))
    r = subprocess.run([sys.executable, '-c', CODE], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
    print('%-24s %s' % (spec, line[0][7:] if line else r.stderr[-300:]), flush=True)
