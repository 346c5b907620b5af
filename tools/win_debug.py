#!/usr/bin/env python3
"""Error pattern of the windowed attention kernels against an fp64 CPU evaluation (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
from oracle import sam as osam
DEV = "cuda:0"; BF16 = torch.bfloat16
B, H, grid = 1, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 28
torch.manual_seed(0)
d = H * 64
qkv = torch.randn((B * grid * grid, 3 * d), device=DEV).to(BF16)
pad = torch.randn((3 * d,), device=DEV).to(BF16)
rh = torch.randn((27, 64), device=DEV) * 0.3; rw = torch.randn((27, 64), device=DEV) * 0.3
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
if "nobias" in mode: rh.zero_(); rw.zero_()
if "noh" in mode: rh.zero_()
if "now" in mode: rw.zero_()
if "q0" in mode: qkv.view(-1, 3, d)[:, 0] = 0; pad.view(3, d)[0] = 0
if "v1" in mode: qkv.view(-1, 3, d)[:, 2] = 1; pad.view(3, d)[2] = 1
print("MODE", mode)
new = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=torch.float32).cpu()
old = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=torch.float32, variant=1).cpu()
nW = (grid + 13) // 14; P = nW * 14
x = pad.double().cpu().repeat(B, P, P, 1); x[:, :grid, :grid] = qkv.double().cpu().view(B, grid, grid, 3 * d)
xw = x.view(B, nW, 14, nW, 14, 3, H, 64).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nW * nW, 3, H, 196, 64)
q, k, v = (xw[:, i].reshape(-1, 196, 64) for i in range(3))
bias = osam.rel_pos_bias(q, rh.to(BF16).double().cpu(), rw.to(BF16).double().cpu(), 14)
att = (torch.softmax((q * 0.125) @ k.transpose(1, 2) + bias, dim=-1) @ v).reshape(-1, H, 196, 64)
ref = att.reshape(B, nW, nW, H, 14, 14, 64).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, P, P, d)[:, :grid, :grid].reshape(-1, d).float()
for name, got in (("old", old), ("new", new)):
    err = (got - ref).abs().view(B, grid, grid, H, 64)
    print(name, "max", float(err.max()), "rel_l2", float((got - ref).norm() / ref.norm()))
    bad = err > 0.05
    print("  frac bad", float(bad.float().mean()))
    print("  by head", bad.float().mean(dim=(0, 1, 2, 4)).tolist())
    print("  by dim block(8)", [round(float(v), 3) for v in bad.float().mean(dim=(0, 1, 2, 3)).view(8, 8).mean(1)])
    yy = bad.float().mean(dim=(0, 2, 3, 4)); xx = bad.float().mean(dim=(0, 1, 3, 4))
    print("  by y", [round(float(v), 2) for v in yy])
    print("  by x", [round(float(v), 2) for v in xx])
