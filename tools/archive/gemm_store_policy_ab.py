#!/usr/bin/env python3
"""A/B of the C-store cache policy of the persistent GEMM (default / nt / sc1; COR_PROBES build, cfg bits 21 / 22) at the SAM-B block shapes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
_native.use_probe_library()
CF = {"default": 13, "nt": 13 | (0x2000 << 8), "sc1": 13 | (0x4000 << 8), "nt_res_loads": 13 | (0x8000 << 8), "nt_both": 13 | (0xA000 << 8)}
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
    ts = {k: [] for k in CF}
    for r in range(7):
        for k, cfg in CF.items():
            if mode == "res": x.normal_()
            for _ in range(2): run(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(cfg)
            e1.record(); e1.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps(dict(shape=f"{M}x{N}x{K} {name}", **{k + "_us": round(med(v), 1) for k, v in ts.items()})), flush=True)
