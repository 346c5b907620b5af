#!/usr/bin/env python3
"""gemm_fr (cfg 15, free-running 8-wave form) against gemm_pp (cfg 13): bit equality on block shapes + ragged edges, then timing.
Needs profiles/r03_gemm_free_running_negative.patch applied (the shipped library answers cfg 15 with COR_EINVAL): the kernel was
measured slower and is kept as a patch + this checker only."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
T = torch.bfloat16
ok = True
for (M, N, K) in [(131072, 2304, 768), (131072, 768, 768), (131072, 3072, 768), (131072, 768, 3072), (65536 + 77, 1024 + 8, 192), (70000, 776, 256), (16384 * 3 + 5, 2304, 320)]:
    A = torch.randn((M, K), device="cuda").to(T); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(T); b = torch.randn((N,), device="cuda")
    for mode in ("plain", "gelu", "res"):
        outs = []
        for cfg in (13, 15):
            if mode == "res":
                x = torch.ones((M, N), device="cuda") * 0.5
                outs.append(ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x, cfg=cfg).clone())
            else:
                outs.append(ops.gemm(A, W, out_dtype=T, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg).clone())
        eq = torch.equal(outs[0], outs[1]); ok &= eq
        print(json.dumps(dict(M=M, N=N, K=K, mode=mode, bit_equal=eq)), flush=True)
    del A, W
print("ALL EQUAL" if ok else "MISMATCH", flush=True)
M = 131072
for N, K, mode, name in [(2304, 768, "plain", "qkv"), (768, 768, "res", "proj+res"), (3072, 768, "gelu", "lin1+gelu"), (768, 3072, "res", "lin2+res")]:
    A = torch.randn((M, K), device="cuda").to(T); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(T); b = torch.randn((N,), device="cuda")
    x = torch.randn((M, N), device="cuda") if mode == "res" else None
    def run(cfg):
        if mode == "res": ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x, cfg=cfg)
        else: ops.gemm(A, W, out_dtype=T, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg)
    ts = {13: [], 15: []}
    for r in range(7):
        for cfg in (13, 15):
            for _ in range(2): run(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(cfg)
            e1.record(); e1.synchronize(); ts[cfg].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps(dict(shape=name, pp_us=round(med(ts[13]), 1), fr_us=round(med(ts[15]), 1))), flush=True)
