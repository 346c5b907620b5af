"""Top-level CORE model on the HIP engine. Same constructor / forward signature / return tuple / state_dict keys as
the reference's CirSegModelWithQuerySupportFeat (lib/sam_with_sup_branch.py:19-104)."""
from typing import Any, List

import torch
from torch import nn

from .. import engine
from .sam_model.image_encoder import ImageEncoderViT
from .sam_model.mask_decoder import MaskDecoder
from .sam_model.my_prompt_encoder import PromptEncoder
from .support_branch import SupportBranch


class CirSegModelWithQuerySupportFeat(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder: ImageEncoderViT, prompt_encoder: PromptEncoder, support_branch: SupportBranch,
                 mask_decoder: MaskDecoder, pixel_mean: List[float] = [123.675, 116.28, 103.53],
                 pixel_std: List[float] = [58.395, 57.12, 57.375]) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.support_branch = support_branch
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.compute_dtype = torch.float32      # torch.float32: exact-fp32 MFMA ; torch.bfloat16: fast mode
        self._packed = {}

    @property
    def device(self) -> Any:
        return self.pixel_mean.device

    # ---- packed-weight cache: rebuilt after anything that can change parameters or their device
    def invalidate_packed(self):
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        return super().load_state_dict(*a, **k)

    def _resolve_dtype(self):
        if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16      # the reference runs inference under accelerator.autocast() bf16 (vaild_a.yaml:4)
        return self.compute_dtype

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("cor_amd: the model must live on a GPU (model.to('cuda')); there is no CPU path")

    def _fingerprint(self):
        """Cheap change detector for the packed-weight cache: every in-place update (optimizer step, p.data.copy_, a
        submodule's load_state_dict) bumps the tensor's _version; re-allocation (.to(), .half()) changes data_ptr."""
        return hash(tuple((t._version, t.data_ptr()) for t in list(self.parameters()) + list(self.buffers())))

    def packed(self, T=None):
        T = T or self._resolve_dtype()
        fp = self._fingerprint()
        if self._packed.get("fp") != fp:
            self._packed = {"fp": fp}
        if T not in self._packed:
            with torch.no_grad():
                self._packed[T] = engine.pack(self.state_dict(), self.image_encoder.cfg, self.support_branch.siglip.cfg,
                                              self.support_branch.mask_pooling_name, T)
        return self._packed[T]

    @torch.no_grad()
    def forward(self, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs, multimask_output=True):
        """-> (final_masks f32[B,1,256,256] logits, query_image_embeddings f32[B,256,64,64], comb_support_feat f32[B,1,256])"""
        if self.training:
            raise RuntimeError("cor_amd implements the retrieval-time (inference) forward only: call model.eval() first "
                               "(training / backward are out of scope)")
        T = self._resolve_dtype()
        self._require_gpu()
        with torch.cuda.device(self.device):       # the kernels launch on the CURRENT device: make it the model's
            return engine.forward(self.packed(T), self.image_encoder.cfg, self.support_branch.siglip.cfg,
                                  self.support_branch.mask_pooling_name, T, query_image_inputs, support_image_inputs,
                                  change_text_inputs, support_mask_inputs, multimask_output)

    @torch.no_grad()
    def forward_with_aux(self, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs,
                         multimask_output=True):
        """forward() plus {'masks': all 4 mask logits, 'iou': [B,4], 'best': [B]} for parity tests / analysis."""
        T = self._resolve_dtype()
        self._require_gpu()
        with torch.cuda.device(self.device):
            return engine.forward(self.packed(T), self.image_encoder.cfg, self.support_branch.siglip.cfg,
                                  self.support_branch.mask_pooling_name, T, query_image_inputs, support_image_inputs,
                                  change_text_inputs, support_mask_inputs, multimask_output, return_aux=True)
