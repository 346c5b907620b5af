#!/usr/bin/env python3
"""Yardstick (not a product path): the vendor libraries reached through torch (hipBLASLt / rocBLAS GEMM, the SDPA flash kernel) on the
batch-32 SAM-B block shapes, beside the hand-written kernels WITH their fused epilogues. The library numbers are for the bare
product (bf16 C = A·W^T [+ bias]); the residual add / GELU / fp32 output the hand-written epilogue does would be further kernels.
Output: JSON lines (profiles/archive/r03_library_yardstick.jsonl)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from cor_amd import ops


def timed(fn, n=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


M = 131072
for N, K, mode, name in [(2304, 768, "plain", "qkv"), (768, 768, "res", "proj+res"), (3072, 768, "gelu", "lin1+gelu"), (768, 3072, "res", "lin2+res")]:
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
    b32 = torch.randn((N,), device="cuda"); b16 = b32.to(torch.bfloat16)
    x = torch.randn((M, N), device="cuda") if mode == "res" else None
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    rec = dict(kind="gemm", shape=f"{M}x{N}x{K} {name}", gflop=2e-9 * M * N * K)
    rec["lib_mm_us"] = timed(lambda: torch.mm(A, W.t(), out=out))
    rec["lib_linear_bias_us"] = timed(lambda: F.linear(A, W, b16))
    if mode == "res":
        def lib_full():
            y = F.linear(A, W, b16)
            x.add_(y)
        rec["lib_linear_plus_residual_add_us"] = timed(lib_full)
        rec["hip_fused_us"] = timed(lambda: ops.gemm(A, W, out_dtype=torch.float32, bias=b32, residual=x, out=x))
    elif mode == "gelu":
        rec["lib_linear_plus_gelu_us"] = timed(lambda: F.gelu(F.linear(A, W, b16)))
        rec["hip_fused_us"] = timed(lambda: ops.gemm(A, W, out_dtype=torch.bfloat16, bias=b32, act=ops.ACT_GELU_ERF))
    else:
        rec["hip_fused_us"] = timed(lambda: ops.gemm(A, W, out_dtype=torch.bfloat16, bias=b32))
    for k in list(rec):
        if k.endswith("_us"):
            rec[k.replace("_us", "_tflops")] = round(rec["gflop"] / rec[k] * 1e3, 1)     # GFLOP / us = PFLOP/s
            rec[k] = round(rec[k], 1)
    print(json.dumps(rec), flush=True)
    del A, W, x, out
    torch.cuda.empty_cache()

# global attention, B=32 x 12 heads x 4096 x 64, WITHOUT the decomposed relative-position bias (the library kernel has no such input)
B, H, S, D = 32, 12, 4096, 64
q, k, v = (torch.randn((B, H, S, D), device="cuda").to(torch.bfloat16) for _ in range(3))
rec = dict(kind="attention", shape=f"B{B} H{H} S{S} D{D}", gflop=4e-9 * B * H * S * S * D)
for name, be in [("flash", "FLASH_ATTENTION"), ("efficient", "EFFICIENT_ATTENTION")]:
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        with sdpa_kernel(getattr(SDPBackend, be)):
            rec[f"lib_sdpa_{name}_nobias_us"] = round(timed(lambda: F.scaled_dot_product_attention(q, k, v), n=6, warm=2), 1)
    except Exception as e:  # noqa: BLE001
        rec[f"lib_sdpa_{name}_error"] = str(e)[:120]
print(json.dumps(rec), flush=True)
