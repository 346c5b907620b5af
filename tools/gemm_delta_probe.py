#!/usr/bin/env python3
"""VERDICT r4 item 2a, measured before building it into the engine: does moving the residual add out of the proj GEMM's fp32
epilogue (bf16 "delta" output) into the LayerNorm pass pay? Same box, one process, alternating, batch-32 SAM-B shapes.
  A (shipped):  x = x + proj(a) [fp32 in-place residual epilogue]            ; h = LN(x)
  B (delta, x written by the LN pass):  d = proj(a) [bf16 out]               ; x += d, h = LN(x)   (one pass: reads x + d, writes x + h)
  C (delta, x NOT written; lin2's epilogue would take d as a second residual): d = proj(a) ; h = LN(x + d)   (lower bound of that form)
usage (GPU box): COR_AMD_LIB=tools/probes/libcor_probes.so python tools/gemm_delta_probe.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cor_amd import ops, _native as nat
dev, BF, F32 = "cuda:0", torch.bfloat16, torch.float32
lib = nat.load()
lib.cor_probe_layernorm_delta.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
M, D = 131072, 768
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn((M, D), device=dev, generator=g).to(BF)
w = (torch.randn((D, D), device=dev, generator=g) * 0.03).to(BF)
bias = torch.randn((D,), device=dev, generator=g)
x = torch.randn((M, D), device=dev, generator=g)
lw, lb = torch.randn((D,), device=dev, generator=g), torch.randn((D,), device=dev, generator=g)
h = torch.empty((M, D), device=dev, dtype=BF)
d = torch.empty((M, D), device=dev, dtype=BF)
s = torch.cuda.current_stream().cuda_stream

def A():
    ops.gemm(a, w, out_dtype=F32, bias=bias, residual=x, out=x)
    ops.layernorm(x, lw, lb, 1e-6, out=h, reverse=True)
def B():
    ops.gemm(a, w, out_dtype=BF, bias=bias, out=d)
    nat.check(lib.cor_probe_layernorm_delta(x.data_ptr(), d.data_ptr(), h.data_ptr(), lw.data_ptr(), lb.data_ptr(), M, D, 1e-6, 1, s), "probe")
def C():
    ops.gemm(a, w, out_dtype=BF, bias=bias, out=d)
    nat.check(lib.cor_probe_layernorm_delta(x.data_ptr(), d.data_ptr(), h.data_ptr(), lw.data_ptr(), lb.data_ptr(), M, D, 1e-6, 0, s), "probe")
def Ag(): ops.gemm(a, w, out_dtype=F32, bias=bias, residual=x, out=x)
def Al(): ops.layernorm(x, lw, lb, 1e-6, out=h, reverse=True)
def Bg(): ops.gemm(a, w, out_dtype=BF, bias=bias, out=d)
def Bl(): nat.check(lib.cor_probe_layernorm_delta(x.data_ptr(), d.data_ptr(), h.data_ptr(), lw.data_ptr(), lb.data_ptr(), M, D, 1e-6, 1, s), "probe")
def Cl(): nat.check(lib.cor_probe_layernorm_delta(x.data_ptr(), d.data_ptr(), h.data_ptr(), lw.data_ptr(), lb.data_ptr(), M, D, 1e-6, 0, s), "probe")

# check B against A once
x0 = x.clone(); A(); hA, xA = h.clone(), x.clone(); x.copy_(x0); B(); torch.cuda.synchronize()
print(json.dumps(dict(check="B vs A", x_max_abs=float((x - xA).abs().max()), h_max_abs=float((h.float() - hA.float()).abs().max()), x_scale=float(xA.abs().max()))), flush=True)
x.copy_(x0)
def timeit(f, n=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(3):
    r = {k: round(timeit(f), 1) for k, f in (("A_pair", A), ("B_pair", B), ("C_pair", C), ("A_gemm_res_f32", Ag), ("A_ln", Al), ("B_gemm_bf16", Bg), ("B_ln_delta_write_x", Bl), ("C_ln_delta_no_x", Cl))}
    x.copy_(x0)
    print(json.dumps(dict(round=rnd, unit="us", **r)), flush=True)
