#!/usr/bin/env python3
"""Cycle stamps of block 0 / thread 0 of gemm_pp (probe build): where a tile's time goes.
usage: python tools/dbg/gemm_stamps.py [M N K [extra_dbg_hex]]   (fp32 out + residual)
stamps per tile: 0 tile start, 1 K loop done, 2 rows realigned, 3 bias landed, 4 next prologue issued, 5..12 blocks written, 15 end"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
_native.use_probe_library()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (131072, 768, 768)
extra = int(sys.argv[4], 16) if len(sys.argv) > 4 else 0
mode = sys.argv[5] if len(sys.argv) > 5 else "res"          # res: fp32 out + residual | bf16: bf16 out, bias | gelu: bf16 out, bias + GELU
dev = "cuda:0"
A = torch.randn((M, K), device=dev).bfloat16(); W = (torch.randn((N, K), device=dev) / K ** 0.5).bfloat16()
bias = torch.randn((N,), device=dev); R = torch.randn((M, N), device=dev)
st = torch.zeros((max(N, 1024),), device=dev, dtype=torch.float32)   # 16 stamps x 8 bytes x (N / 32) tiles
cfg = 13 | ((0x100000 | extra) << 8)
for _ in range(3):
    st.zero_()
    if mode == "res": ops.gemm(A, W, out_dtype=torch.float32, bias=bias, residual=R, col_scale=st[:N], cfg=cfg)
    else: ops.gemm(A, W, out_dtype=torch.bfloat16, bias=bias, act=1 if mode == "gelu" else 0, col_scale=st[:N], cfg=cfg)
torch.cuda.synchronize()
t = st.view(torch.int64)[:16 * 32].cpu().view(-1, 16)
names = ["start", "kloop", "realign", "bias", "pre"] + [f"blk{i}" for i in range(8)] + ["", "", "end"]
rows_ = [r for r in t.tolist() if r[0]]
if rows_ and rows_[0][13] and rows_[-1][14]:
    dc, dr = rows_[-1][15] - rows_[0][0], rows_[-1][14] - rows_[0][13]
    print(f"in-kernel clock: {dc} cycles (s_memtime) in {dr} ticks of s_memrealtime (100 MHz) = {dc / dr * 0.1:.3f} GHz over {dr / 100:.1f} us, {len(rows_)} tiles")
for ti in range(t.shape[0]):
    row = t[ti].tolist()
    if row[0] == 0: continue
    row[13] = row[14] = 0
    base = row[0]
    print(f"tile {ti}: " + " ".join(f"{names[i]}+{(row[i] - (row[i - 1] if i and row[i - 1] else base))}" for i in range(16) if row[i]))
    print(f"        total {row[15] - base} counter units; K loop {row[1] - base}, epilogue {row[15] - row[2]}")
