#!/usr/bin/env python3
"""In-kernel cycle breakdown of the global SAM attention kernel (variant 9 = timing-probe build of flash_global_pipe: every wave
sums s_memtime deltas per loop section over its 64 key tiles and writes them INSTEAD of its outputs). python tools/attn_stamps.py [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native
_native.use_probe_library()     # variant 9 (cycle stamps instead of outputs) exists in the COR_PROBES build only: make -C cor_amd/csrc probes
from cor_amd._native import Q_PRESCALE_HD64 as QC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, g, dev, T = 12, 64, "cuda:0", torch.bfloat16
d = H * 64
qkv = torch.randn((B * g * g, 3 * d), device=dev).to(T)
pad = torch.randn((3 * d,), device=dev).to(T)
rh = torch.randn((127, 64), device=dev) * 0.5
rw = torch.randn((127, 64), device=dev) * 0.5
for _ in range(30):
    ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0, q_prescale=QC)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); out = ops.sam_attention(qkv, pad, rh, rw, B, H, g, 0, q_prescale=QC, variant=9); e1.record(); torch.cuda.synchronize()
nw = 4
nblk = (g * g // (32 * nw)) * H * B
st = out.view(torch.int64).flatten()[: nblk * nw * 8].view(nblk * nw, 8)[:, :6].double()
names = ["dma_issue", "phase_A(PV+max)", "ref_update", "phase_B(QK+exp)", "vmcnt_wait", "barrier"]
per_tile = st.mean(0) / 64.0
tot = float(per_tile.sum())
print(json.dumps(dict(kernel_ms_probe=e0.elapsed_time(e1), cycles_per_tile_total=tot,
                      **{n: round(float(v), 1) for n, v in zip(names, per_tile)},
                      p10={n: round(float(v) / 64, 1) for n, v in zip(names, st.quantile(0.1, dim=0))},
                      p90={n: round(float(v) / 64, 1) for n, v in zip(names, st.quantile(0.9, dim=0))})))
