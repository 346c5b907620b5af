#!/usr/bin/env python3
"""Small-M GEMM shapes (text tower at batch 1..4: M = 64..256; decoder token rows: M = 6..192) under every shipped tile kernel:
time per launch and bit-equality with the automatic choice. python tools/gemm_small_m.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cor_amd import ops
BF, F32 = torch.bfloat16, torch.float32
def t(f, n=50):
    """device time per launch inside a replayed hipGraph of n launches (eager launches are host-bound at ~13 us each)"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = ((4096, 2304, 768, "plain"), (4096, 768, 768, "res"), (4096, 3072, 768, "gelu"), (4096, 768, 3072, "res"), (8192, 2304, 768, "plain"), (8192, 768, 768, "res"), (8192, 3072, 768, "gelu"), (8192, 768, 3072, "res"),
          (16384, 768, 768, "res"), (16384, 768, 3072, "res"), (1152, 768, 768, "res"), (1152, 768, 3072, "res"), (1152, 2304, 768, "plain"), (2304, 768, 3072, "res")) if len(sys.argv) > 1 and sys.argv[1] == "mid" else None
if len(sys.argv) > 1 and sys.argv[1] == "b32":   # the batch-32 step's inefficient launches (profiles/r05_gemm_shapes.jsonl)
    SHAPES = ((2048, 768, 3072, "res"), (2048, 3072, 768, "gelu"), (2048, 2304, 768, "plain"), (2048, 768, 768, "res"), (18432, 768, 768, "res"), (18432, 768, 3072, "res"),
              (18432, 2304, 768, "plain"), (18432, 3072, 768, "gelu"), (131072, 128, 256, "plain"), (131072, 256, 128, "res"), (18432, 1024, 256, "gelu"), (18432, 256, 1024, "res"))
for (M, N, K, mode) in SHAPES or ((64, 2304, 768, "plain"), (64, 768, 768, "res"), (64, 3072, 768, "gelu"), (64, 768, 3072, "res"), (128, 2304, 768, "plain"), (256, 768, 3072, "res"),
                        (6, 256, 256, "plain"), (6, 256, 2048, "res"), (6, 2048, 256, "gelu"), (192, 256, 2048, "res"), (192, 256, 256, "plain"), (576, 2304, 768, "plain"), (576, 768, 3072, "res")):
    A = torch.randn((M, K), device="cuda").to(BF); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(BF)
    b = torch.randn((N,), device="cuda"); x0 = torch.randn((M, N), device="cuda")
    def run(cfg):
        if mode == "res":
            x = x0.clone(); ops.gemm(A, W, out_dtype=F32, bias=b, residual=x, out=x, cfg=cfg); return x
        return ops.gemm(A, W, out_dtype=BF, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg)
    ref = run(0)
    rec = dict(M=M, N=N, K=K, mode=mode)
    for cfg in (0, 1, 2, 3, 4, 13):
        try:
            out = run(cfg)
            x = x0.clone()
            f = (lambda: ops.gemm(A, W, out_dtype=F32, bias=b, residual=x, out=x, cfg=cfg)) if mode == "res" else (lambda: ops.gemm(A, W, out_dtype=BF, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE, cfg=cfg))
            rec[f"cfg{cfg}_us"] = round(t(f), 1); rec[f"cfg{cfg}_bitwise"] = bool(torch.equal(out, ref))
        except Exception as e:
            rec[f"cfg{cfg}_us"] = None
    print(json.dumps(rec), flush=True)
