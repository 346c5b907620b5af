"""Factory with the reference's name, signature, accepted values and error behaviour (lib/build_model.py:14-122)."""
import torch

from .. import config
from .sam_model.image_encoder import ImageEncoderViT
from .sam_model.mask_decoder import MaskDecoder
from .sam_model.my_prompt_encoder import PromptEncoder
from .sam_model.transformer import TwoWayTransformer
from .sam_with_sup_branch import CirSegModelWithQuerySupportFeat
from .support_branch import SupportBranch


def build_model_with_query_support_feat(sam_model="sam_base", siglip_model="ViT-SO400M-14-SigLIP-384", sam_checkpoint_path=None,
                                        siglip_checkpoint_path=None, mask_pooling="MaskedPooling"):
    sc = config.sam_cfg(sam_model)          # raises ValueError("Invalid SAM model: ...") like the reference (:49)
    grid = sc["img"] // sc["patch"]
    model = CirSegModelWithQuerySupportFeat(
        image_encoder=ImageEncoderViT(img_size=sc["img"], patch_size=sc["patch"], embed_dim=sc["dim"], depth=sc["depth"],
                                      num_heads=sc["heads"], out_chans=sc["out"], window_size=sc["window"],
                                      global_attn_indexes=sc["global_idx"]),
        support_branch=SupportBranch(clip_model=siglip_model, siglip_path=siglip_checkpoint_path, mask_pooling=mask_pooling),
        prompt_encoder=PromptEncoder(embed_dim=sc["out"], image_embedding_size=(grid, grid)),
        mask_decoder=MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=sc["out"], mlp_dim=2048, num_heads=8),
                                 transformer_dim=sc["out"], iou_head_depth=3, iou_head_hidden_dim=256),
    )
    if sam_checkpoint_path is not None:     # :96-119 — same key-prefix split, strict=False, same freezes
        sd = torch.load(sam_checkpoint_path, map_location="cpu")
        pick = lambda pre, match=None: {k[len(pre):]: v for k, v in sd.items() if k.startswith(match or pre)}
        model.image_encoder.load_state_dict(pick("image_encoder."), strict=False)
        model.mask_decoder.load_state_dict(pick("mask_decoder."), strict=False)
        model.prompt_encoder.load_state_dict(pick("prompt_encoder.", "prompt_encoder.dense_embedding"), strict=False)
        print(f"Load SAM Checkpoint: {sam_checkpoint_path}.")
        model.support_branch.siglip.freeze()
        model.image_encoder.freeze()
        print("Freeze weight of SAM_Image_Encoder and SigLIP.")
        for p in model.mask_decoder.iou_prediction_head.parameters():
            p.requires_grad = False
        print("Freeze weight of mask_decoder.iou_prediction_head.")
        model.invalidate_packed()
    return model
