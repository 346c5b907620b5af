import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cor_amd import ops, _native
lib = _native.load(); dev = "cuda:0"; T = torch.bfloat16
B, H, g = 2, 12, 64
d = H * 64
torch.manual_seed(0)
qkv = torch.randn((B * g * g, 3 * d), device=dev).to(T)
pad = torch.randn((3 * d,), device=dev).to(T)
rh = torch.randn((127, 64), device=dev) * 0.5; rw = torch.randn((127, 64), device=dev) * 0.5
o0 = ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0, variant=1).float()   # chain form
o1 = ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0).float()              # default (pipelined)
torch.cuda.synchronize()
print("maxdiff", float((o0 - o1).abs().max()), "ref max", float(o0.abs().max()), "finite", bool(torch.isfinite(o1).all()))
