#!/usr/bin/env python3
"""Kernel mix of one training configuration under rocprofv3 (measurement aid):
   rocprofv3 --kernel-trace --stats ... -- python tools/prof_train.py [dopri5|euler]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_train

if __name__ == '__main__':
    for m in (sys.argv[1:] or ['dopri5']):
        print(bench_train.one_case('100k-node grid, H=256, %s' % m, 316, 256, 10, m, torch.device('cuda:0'), 0))
