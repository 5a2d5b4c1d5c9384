#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | head -5
python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --sharded --no-cpu-baseline 2>&1 | grep -E '^\{"metric' | grep -o '"ms_per_step": [0-9.]*'
} > gpurun_out/exp_sharded.log 2>&1
