#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc). python tools/gemm_one.py M N K [cfg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
M, N, K = (int(v) for v in sys.argv[1:4]); cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn((M, K), device="cuda").to(torch.bfloat16); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn((N,), device="cuda")
for _ in range(5): ops.gemm(A, W, out_dtype=torch.bfloat16, bias=b, cfg=cfg)
torch.cuda.synchronize()
