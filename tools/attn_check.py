import sys, torch
sys.path.insert(0, "/root/repo")
from cor_amd import ops, _native
lib = _native.load(); dev = "cuda:0"; T = torch.bfloat16
B, H, g = 2, 12, 64
d = H * 64
torch.manual_seed(0)
qkv = torch.randn((B * g * g, 3 * d), device=dev).to(T)
pad = torch.randn((3 * d,), device=dev).to(T)
rh = torch.randn((127, 64), device=dev) * 0.5; rw = torch.randn((127, 64), device=dev) * 0.5
lib.cor_flash_set_variant(0); o0 = ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0).float()
lib.cor_flash_set_variant(1); o1 = ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0).float()
torch.cuda.synchronize()
print("maxdiff", float((o0 - o1).abs().max()), "ref max", float(o0.abs().max()), "finite", bool(torch.isfinite(o1).all()))
