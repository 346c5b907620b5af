"""Oracle (test infrastructure): top-level CORE forward.

ref: lib/sam_with_sup_branch.py:57-104 (CirSegModelWithQuerySupportFeat.forward).
"""
from __future__ import annotations

import torch

from . import config, sam, support


def forward(sd, sam_model, siglip_model, mask_pooling,
            query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs,
            multimask_output=True, return_aux=False):
    """-> (final_masks [B,1,256,256] logits, query_image_embeddings [B,256,64,64], comb_support_feat [B,1,256])"""
    scfg, gcfg = config.sam_cfg(sam_model), config.siglip_cfg(siglip_model)
    B = query_image_inputs.shape[0]
    emb = sam.image_encoder(sd, query_image_inputs, scfg)                                  # :76
    feat = support.support_branch(sd, support_image_inputs, change_text_inputs,
                                  support_mask_inputs, gcfg, mask_pooling)                 # :79
    dense = sam.dense_no_mask(sd, B)                                                       # :82
    masks, iou, _ = sam.mask_decoder(sd, emb, sam.dense_pe(sd), feat, dense, multimask_output)  # :86-92
    if multimask_output:                                                                   # :96-100
        best = iou.argmax(dim=1)
        final = masks[torch.arange(B), best].unsqueeze(1)
    else:
        best = torch.zeros(B, dtype=torch.long)
        final = masks
    if return_aux:
        return final, emb, feat, dict(masks=masks, iou=iou, best=best)
    return final, emb, feat
