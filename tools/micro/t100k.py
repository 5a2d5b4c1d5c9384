import sys, os, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tools')
import bench_train
print(bench_train.one_case('100k', 316, 256, 10, 'dopri5', torch.device('cuda:0'), 0)['gpu_ms_per_adam_step'])
