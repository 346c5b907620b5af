#!/usr/bin/env python3
"""Throughput of the device pre-processing (one query image 1024^2, one support image 384^2 and one support mask per triplet)
from 640x480 uint8 sources: python tools/preprocess_bench.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import preprocess as CP
dev = "cuda:0"
q = torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device=dev)
m = torch.randint(0, 256, (480, 640), dtype=torch.uint8, device=dev)
tq, ts, tm = CP.QueryImageTransform(), CP.ImageTransform(384), CP.MaskTransform(384)
def triplet():
    return tq(q), ts(q), tm(m)
for _ in range(5): triplet()
torch.cuda.synchronize()
N = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): triplet()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / N
# algorithmic bytes per triplet: sources read once, uint8 intermediates written + read, float32 outputs written
b = (480 * 640 * 3 + 480 * 1024 * 3 * 2 + 3 * 1024 * 1024 * 4) + (480 * 640 * 3 + 480 * 384 * 3 * 2 + 3 * 384 * 384 * 4) + (480 * 640 + 480 * 384 * 2 + 384 * 384 * 4)
print(json.dumps(dict(ms_per_triplet=ms, triplets_per_s=1e3 / ms, algorithmic_MB=b / 1e6, GBps=b / ms / 1e6,
                      note="6 kernel launches per triplet from Python; HBM-bound byte work, launch-bound at this size")))
