import sys, collections, torch
sys.path.insert(0, "/root/repo")
from cor_amd import ops, utils
from cor_amd.lib.build_model import build_model_with_query_support_feat
import cor_amd.ops as O
shapes = collections.Counter()
orig = O.gemm
def spy(a, w, out_dtype=None, **kw):
    M = a.shape[0]; K = a.shape[1]; N = w.shape[0]
    shapes[(M, N, K, str(a.dtype).split('.')[-1], str(out_dtype).split('.')[-1], kw.get('act', 0), kw.get('residual') is not None)] += 1
    return orig(a, w, out_dtype=out_dtype, **kw)
O.gemm = spy
import cor_amd.engine as E
model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
utils.randomize_parameters(model, seed=0); model = model.to("cuda").eval(); model.compute_dtype = torch.bfloat16
b = utils.synthetic_batch(32, "cuda", seed=0)
model(**b, multimask_output=True)
for k, v in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
    M, N, K = k[:3]; t = ((M + 255) // 256) * ((N + 255) // 256)
    print(v, k, "tiles256", t, "GF", round(2 * M * N * K * v / 1e9, 1))
