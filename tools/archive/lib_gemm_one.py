#!/usr/bin/env python3
"""Yardstick only: the vendor library's GEMM (torch F.linear -> hipBLASLt) on one shape, a few launches, for rocprofv3 --pmc passes
beside tools/gemm_one.py. python tools/lib_gemm_one.py M N K"""
import sys
import torch
import torch.nn.functional as F
M, N, K = (int(v) for v in sys.argv[1:4])
A = torch.randn((M, K), device="cuda").to(torch.bfloat16); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn((N,), device="cuda").to(torch.bfloat16)
for _ in range(5):
    y = F.linear(A, W, b)
torch.cuda.synchronize()
