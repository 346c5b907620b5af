#!/usr/bin/env python3
"""A/B of the persistent GEMM's tile orders (XCD-stationary W-panels vs the round-2 banded order) on the SAM-B block shapes at
batch 32, interleaved rounds in one process, COR_PROBES build (make -C cor_amd/csrc probes). python tools/gemm_order_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
_native.use_probe_library()
OLD = 13 | (1 << 20)
M, dev, T = 131072, "cuda:0", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 2304, 768, "plain"), ("proj+res", 768, 768, "res"), ("lin1+gelu", 3072, 768, "gelu"), ("lin2+res", 768, 3072, "res"),
          ("neck 3x3", 256, 2304, "plain32"), ("siglip fc1", 3072, 768, "gelu_s")]
for name, N, K, mode in shapes:
    Mm = 18432 if mode.endswith("_s") else M
    A = torch.randn((Mm, K), generator=g, device=dev).to(T); W = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(T)
    b = torch.randn((N,), generator=g, device=dev)
    x = torch.randn((Mm, N), generator=g, device=dev) if mode == "res" else None
    def run(cfg):
        if mode == "res":
            return ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x, cfg=cfg)
        return ops.gemm(A, W, out_dtype=torch.float32 if mode == "plain32" else T, bias=b, act=ops.ACT_GELU_ERF if mode.startswith("gelu") else ops.ACT_NONE, cfg=cfg)
    if mode != "res":
        assert torch.equal(run(13), run(OLD)), name            # the order never changes a result
    ts = {13: [], OLD: []}
    for r in range(7):
        for cfg in (13, OLD):
            if mode == "res": x.normal_()
            for _ in range(2): run(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(cfg)
            e1.record(); e1.synchronize()
            ts[cfg].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps(dict(shape=f"{Mm}x{N}x{K} {name}", xcd_stationary_us=round(med(ts[13]), 1), banded_us=round(med(ts[OLD]), 1),
                          xcd_stationary_min=round(min(ts[13]), 1), banded_min=round(min(ts[OLD]), 1))), flush=True)
