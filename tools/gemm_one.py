#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc). python tools/gemm_one.py M N K [cfg] [plain|gelu|res]
plain: bf16 out + bias; gelu: bf16 out + bias + erf GELU (lin1); res: fp32 out + bias + in-place fp32 residual (proj / lin2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
if len(sys.argv) > 4 and (int(sys.argv[4]) >> 8 or (int(sys.argv[4]) & 0xff) == 14):      # ablation / order bits, 16x16-MFMA form: only the COR_PROBES build accepts them
    _native.use_probe_library()
M, N, K = (int(v) for v in sys.argv[1:4]); cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
A = torch.randn((M, K), device="cuda").to(torch.bfloat16); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn((N,), device="cuda")
x = torch.randn((M, N), device="cuda") if mode == "res" else None
for _ in range(5):
    if mode == "res":
        ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x, cfg=cfg)
    else:
        ops.gemm(A, W, out_dtype=torch.bfloat16, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg)
torch.cuda.synchronize()
