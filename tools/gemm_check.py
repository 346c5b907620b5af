#!/usr/bin/env python3
"""Full-output comparison of a cor_gemm configuration against configuration 2 (and a torch fp32 product on a sample of rows)
over ragged shapes: python tools/gemm_check.py [cfg=12]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
lib = _native.load(); dev = "cuda:0"; T = torch.bfloat16
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SHAPES = [(4096, 768, 768), (4096 + 40, 1152, 64), (1000, 264, 32), (300, 8, 96), (65536, 2304, 768), (18432, 3072, 768), (131072, 256, 2304),
          (256, 256, 128), (70000, 776, 160)]
bad = 0
for (M, N, K) in SHAPES:
    for od, act, use_res in ((T, 0, False), (T, 1, False), (torch.float32, 0, True), (torch.float32, 0, False)):
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        A = torch.randn((M, K), generator=g, device=dev).to(T)
        W = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(T)
        bias = torch.randn((N,), generator=g, device=dev)
        R = torch.randn((M, N), generator=g, device=dev) if use_res else None
        ref = ops.gemm(A, W, out_dtype=od, bias=bias, act=act, residual=R, cfg=2).float()
        guard = torch.full((M + 2, N), 7.0, device=dev, dtype=od)            # rows before / after must stay untouched
        out = ops.gemm(A, W, out_dtype=od, bias=bias, act=act, residual=R, out=guard[1:M + 1], cfg=cfg)
        torch.cuda.synchronize()
        d = float((out.float() - ref).abs().max())
        ok = d <= (0.07 if od == T else 1e-3) and bool((guard[0] == 7).all()) and bool((guard[M + 1] == 7).all())
        bad += not ok
        print(json.dumps(dict(shape=[M, N, K], out=str(od), act=act, res=use_res, maxdiff=d, ok=ok)), flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
