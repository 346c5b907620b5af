"""cor_amd — MI355X-native (gfx950) implementation of CORE's retrieval-time forward path.

Drop-in surface (same names / signatures as the reference, wangtong627/COR):
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    model = build_model_with_query_support_feat(sam_model=..., siglip_model=..., mask_pooling=...).to("cuda").eval()
    masks, emb, feat = model(query_image_inputs=..., support_image_inputs=..., change_text_inputs=...,
                             support_mask_inputs=..., multimask_output=True)
Retrieval (gallery similarity + top-k, sharded over ranks): cor_amd.retrieval.
All arithmetic runs in hand-written HIP kernels (cor_amd/csrc -> libcor_amd.so); there is no CPU fallback.
"""
__version__ = "0.1.0"
