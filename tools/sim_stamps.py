#!/usr/bin/env python3
"""Cycle stamps of sim_block_scan's sections (COR_PROBES build only: make -C cor_amd/csrc probes). python tools/sim_stamps.py [Bq Ng]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import _native
_native.use_probe_library()
lib = _native.load()
Bq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Ng = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
dev = "cuda:0"
XF = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0      # extra flags: 4 = no candidate passes (timing only)
Q = torch.nn.functional.normalize(torch.randn((Bq, 256), device=dev), dim=-1)
G = torch.nn.functional.normalize(torch.randn((Ng, 256), device=dev), dim=-1).to(torch.bfloat16)
nb = lib.cor_topk_workspace_bytes(Bq, Ng, 10)
ws = torch.zeros((nb + (2 << 20),), dtype=torch.uint8, device=dev)
sc = torch.empty((Bq, 10), dtype=torch.float32, device=dev); ix = torch.empty((Bq, 10), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    rc = lib.cor_similarity_topk(Q.data_ptr(), G.data_ptr(), 1, Bq, Ng, 256, 10, 0, sc.data_ptr(), ix.data_ptr(), ws.data_ptr(), 32 | XF, st)
    assert rc == 0, rc
torch.cuda.synchronize()
import numpy as np
raw = ws.cpu().numpy()
if Ng > 40000 or Bq * 0 + (len(sys.argv) > 3 and sys.argv[3] == "scan"):      # global-threshold pipeline: sim_scan<APPEND> stamps, waves 0 (early) and 4 (late)
    arr = raw[nb: nb + 2 * 24 * 8 * 8].view(np.uint64).reshape(2, 24, 8).astype(np.int64)
    names = ["wait", "barrier", "dma issue", "late epilogue", "mfma tile 0", "epilogue 0", "mfma tile 1", "epilogue 1"]
    for w, label in ((0, "wave 0 (early)"), (1, "wave 4 (late)")):
        a = arr[w]; a = a[a[:, 7] > 0]
        per = np.diff(a[:, 0])                                   # super-tile period
        d = np.diff(a, axis=1)
        print(label, "super-tiles", len(a), "period median", float(np.median(per)) if len(per) else None)
        print("   sections (median cycles):", {names[i + 1]: float(np.median(d[2:, i])) for i in range(7)})
    sys.exit(0)
arr = raw[nb: nb + 8 * 8 * 8 * 4096].view(np.uint64).reshape(-1, 8)
arr = arr[(arr[:, 0] > 0) & (arr[:, 6] > 0)]
d = np.diff(arr[:, :7].astype(np.int64), axis=1)
names = ["dma issue + q convert + frags", "scan", "class maxima + barrier", "tau", "append", "write out"]
print(json.dumps(dict(Bq=Bq, Ng=Ng, waves=int(len(arr)), total_cycles_median=float(np.median(arr[:, 6].astype(np.int64) - arr[:, 0].astype(np.int64))),
                      sections={n: float(np.median(d[:, i])) for i, n in enumerate(names)},
                      sections_max={n: float(np.max(d[:, i])) for i, n in enumerate(names)})))
