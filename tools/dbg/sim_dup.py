import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cor_amd import ops, _native as nat
DEV, BF16 = "cuda:0", torch.bfloat16
rng = np.random.default_rng(6)
Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((40, 256), dtype=np.float32)), dim=-1)
row = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((1, 256), dtype=np.float32)), dim=-1)
G = row.repeat(9000, 1).to(BF16)
G2 = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((9000, 256), dtype=np.float32)), dim=-1)
G2[2000:6000] = torch.nn.functional.normalize(Q[0:1] + 0.05 * row, dim=-1)
G2 = G2.to(BF16)
_, raw2 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK)
_, raw3 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK | nat.TOPK_FORCE_GLOBAL_THRESHOLD)
print("small path flagged:", (raw2 == -2).all(dim=1).nonzero().flatten().tolist())
print("global path flagged:", (raw3 == -2).all(dim=1).nonzero().flatten().tolist())
S = Q.to(BF16).float() @ G2.float().T
for q in (4, 26, 1):
    srt = S[q].sort(descending=True)
    kth = float(srt.values[9])
    print("query", q, "s_dup", float(S[q, 2000]), "kth", kth, "rows >= 0.99 kth:", int((S[q] >= 0.99 * kth).sum()), "rows >= 0.98 kth", int((S[q] >= 0.98 * kth).sum()))
    for sl in range(18):
        lo, hi = sl * 512, min(9000, sl * 512 + 512)
        print("   slice", sl, "max", round(float(S[q, lo:hi].max()), 4), "dups", int(max(0, min(hi, 6000) - max(lo, 2000))))
