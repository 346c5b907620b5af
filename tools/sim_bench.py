#!/usr/bin/env python3
"""GPU micro-benchmark of cor_similarity_topk (gallery similarity + top-k) at the BASELINE shard shapes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
dev = "cuda:0"
for Bq, Ng, k, dt in ((32, 10000, 10, torch.bfloat16), (512, 12500, 10, torch.bfloat16), (512, 125000, 10, torch.bfloat16),
                      (512, 125000, 10, torch.float16), (512, 125000, 10, torch.float32), (512, 1000000, 10, torch.bfloat16)):
    Q = torch.nn.functional.normalize(torch.randn((Bq, 256), device=dev), dim=-1)
    G = torch.nn.functional.normalize(torch.randn((Ng, 256), device=dev), dim=-1).to(dt)
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.similarity_topk(Q, G, k); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts[1:]) * 1e-3
    fl = 2.0 * Bq * Ng * 256
    print(json.dumps(dict(Bq=Bq, Ng=Ng, k=k, dtype=str(dt), us=t * 1e6, tflops=fl / t / 1e12, gallery_GBps=G.numel() * G.element_size() / t / 1e9)), flush=True)
