#!/usr/bin/env python3
"""K sweep of cor_gemm at fixed M,N to separate per-tile fixed cost (prologue+epilogue) from per-K-step cost."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native as _nat
_nat.use_probe_library()        # ablation knobs (bits 8.. of cfg) exist in the COR_PROBES build only: make -C cor_amd/csrc probes, _native
lib = _nat.load(); dev = "cuda:0"; T = torch.bfloat16
cfgs = [int(c) for c in sys.argv[1:]] or [2, 3]   # 7xx = persistent kernel with ablation knob xx (1 no stores, 2 no epilogue, 4 no MFMA)
M, N = 131072, int(os.environ.get("KSWEEP_N", "768"))
for mode in ("bf16_out", "bf16_out_gelu", "f32_out_res"):
    for K in (64, 768, 3072):
        A = torch.randn((M, K), device=dev).to(T); W = (torch.randn((N, K), device=dev) / K ** 0.5).to(T)
        bias = torch.randn((N,), device=dev); R = torch.randn((M, N), device=dev) if mode == "f32_out_res" else None
        od = T if mode.startswith("bf16") else torch.float32
        act = 1 if mode.endswith("gelu") else 0
        row = dict(mode=mode, K=K)
        for c in cfgs:
            kern, knob = (c // 100, c % 100) if c >= 700 else (c, 0)      # 13xx = kernel 13 with ablation knob xx (per-call cfg)
            cfg = kern | (knob << 8)
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(A, W, out_dtype=od, bias=bias, act=act, residual=R, cfg=cfg); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            row[f"cfg{c}_us"] = round(min(ts[1:]) * 1e3, 1)
        print(json.dumps(row), flush=True)
