#!/usr/bin/env python3
"""Where the GEMM time of one bench step goes, by shape and epilogue: HIP events around every cor_gemm launch of a batch-32
bf16 forward (ops.GEMM_PROFILE), aggregated. python tools/gemm_shapes.py [batch]"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, utils
from cor_amd.lib.build_model import build_model_with_query_support_feat
import cor_amd.ops as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
keys = []
orig = O.gemm
def spy(a, w, out_dtype=None, **kw):
    keys.append((a.shape[0], w.shape[0], a.shape[1], str(a.dtype).split('.')[-1], str(out_dtype).split('.')[-1], kw.get('act', 0), kw.get('residual') is not None))
    return orig(a, w, out_dtype=out_dtype, **kw)
import cor_amd.engine as E
E.ops.gemm = spy
model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
utils.randomize_parameters(model, seed=0); model = model.to("cuda").eval(); model.compute_dtype = torch.bfloat16
b = utils.synthetic_batch(B, "cuda", seed=0)
model(**b, multimask_output=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for rep in range(3):
    keys.clear(); O.GEMM_PROFILE = prof = []
    model(**b, multimask_output=True)
    torch.cuda.synchronize(); O.GEMM_PROFILE = None
    for k, p in zip(keys, prof):
        agg[k][0] += 1; agg[k][1] += p[0].elapsed_time(p[1])
tot = sum(v[1] for v in agg.values()) / 3
print(f"GEMM total {tot:.2f} ms per step, {sum(v[0] for v in agg.values()) // 3} launches")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k[:3]
    print(json.dumps(dict(shape=f"{M}x{N}x{K}", ab=k[3], out=k[4], act=k[5], res=k[6], launches=n // 3, ms_per_step=round(ms / 3, 3), share=round(ms / 3 / tot, 3),
                          tflops=round(2.0 * M * N * K * n / (ms * 1e-3) / 1e12, 1))))
