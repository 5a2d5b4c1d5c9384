"""A/B inside ONE box: dopri5 Adam step of a grid case ('100k' or 'M') under environment variants given as arguments 'K=V,K=V' ('-' = defaults)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
case = os.environ.get('AB_CASE', '100k')
side = {'100k': 316, 'M': 1000}[case]
CODE = """
import sys, json, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools')
import bench_train
r = bench_train.one_case(%r, %d, 256, 10, 'dopri5', torch.device('cuda:0'), 0)
print('RESULT', r['gpu_ms_per_adam_step'], r.get('library_kernel_ms_per_step'), r.get('rhs_evaluations_per_step'), 'evals', round(torch.cuda.max_memory_allocated() / 1e9, 2), 'GB',
      {k: v['ms_total'] for k, v in r['breakdown'].items() if v['ms_total'] > 0.02 * r['gpu_ms_per_adam_step']})
""" % (ROOT, ROOT, case, side)
for spec in sys.argv[1:]:
    e = dict(os.environ)
    if spec != '-':
        e.update(dict(kv.split('=') for kv in spec.split(',')))
    r = subprocess.run([sys.executable, '-c', CODE], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
    print('%-44s %s' % (spec, line[0][7:] if line else r.stderr[-400:]), flush=True)
