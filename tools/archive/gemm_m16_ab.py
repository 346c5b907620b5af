#!/usr/bin/env python3
"""A/B of the persistent GEMM on 32x32x16 vs 16x16x32 MFMAs (cfg 13 vs 14; COR_PROBES build), SAM-B block shapes at batch 32."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
_native.use_probe_library()
M, dev, T = 131072, "cuda:0", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, mode in [("qkv", 2304, 768, "plain"), ("proj+res", 768, 768, "res"), ("lin1+gelu", 3072, 768, "gelu"), ("lin2+res", 768, 3072, "res")]:
    A = torch.randn((M, K), generator=g, device=dev).to(T); W = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(T)
    b = torch.randn((N,), generator=g, device=dev)
    x = torch.randn((M, N), generator=g, device=dev) if mode == "res" else None
    def run(cfg):
        if mode == "res":
            return ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x, cfg=cfg)
        return ops.gemm(A, W, out_dtype=T, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg)
    if mode != "res":
        d = (run(13).float() - run(14).float()).abs().max().item()
    ts = {13: [], 14: []}
    for r in range(7):
        for cfg in (13, 14):
            if mode == "res": x.normal_()
            for _ in range(2): run(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(cfg)
            e1.record(); e1.synchronize()
            ts[cfg].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps(dict(shape=f"{M}x{N}x{K} {name}", mfma32_us=round(med(ts[13]), 1), mfma16_us=round(med(ts[14]), 1))), flush=True)
