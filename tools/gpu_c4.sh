#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_odeint.py -m gpu -x -q -k "config_c4 or bench_two" 2>&1 | tail -5 | cut -c1-400
timeout 900 python bench.py --config C4 --no-cpu-baseline 2> gpurun_out/bench_C4.err | grep '^{"metric' > gpurun_out/r02h_bench_C4.json; tail -2 gpurun_out/bench_C4.err | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/r02h_bench_C4.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['spmm_standalone'])"
NDCN_C4_NODES=500000 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 1 --sharded --config C4 --steps 10 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_C4_sharded.err | grep '^{"metric' > gpurun_out/r02h_bench_C4_sharded_1rank.json; tail -2 gpurun_out/bench_C4_sharded.err | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/r02h_bench_C4_sharded_1rank.json')); print(d['ms_per_step'], d['halo_exchange'])"
