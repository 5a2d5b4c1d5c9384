#!/bin/bash
# the multi-GPU code path on the 1-GPU box: tests + bench --sharded with the self-halo hook (RCCL all-to-all-v executes)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "sharded or two_rank or bench_two" > gpurun_out/pytest_sharded.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sharded.log
tail -6 gpurun_out/pytest_sharded.log | cut -c1-300
for h in 0 2000; do
  NDCN_SELF_HALO=$h timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --sharded --steps 10 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_sharded_$h.err | grep '^{"metric' > gpurun_out/bench_sharded_$h.json
  python -c "import json; d=json.load(open('gpurun_out/bench_sharded_$h.json')); print('self_halo $h', d['ms_per_step'], d['halo_exchange'])"
  tail -2 gpurun_out/bench_sharded_$h.err | cut -c1-200
done
