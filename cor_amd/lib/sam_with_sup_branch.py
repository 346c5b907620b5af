"""Top-level CORE model on the HIP engine. Same constructor / forward signature / return tuple / state_dict keys as
the reference's CirSegModelWithQuerySupportFeat (lib/sam_with_sup_branch.py:19-104)."""
import os
import warnings
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
        self._fp_tensors = None                 # _fingerprint(): tensor list of the last module walk, dropped by _apply / load_state_dict

    @property
    def device(self) -> Any:
        return self.pixel_mean.device

    # ---- packed-weight cache: rebuilt after anything that can change parameters or their device
    def invalidate_packed(self):
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self._packed, self._fp_tensors = {}, None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed, self._fp_tensors = {}, None
        return super().load_state_dict(*a, **k)

    def _resolve_dtype(self):
        if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16      # the reference runs inference under accelerator.autocast() bf16 (vaild_a.yaml:4)
        return self.compute_dtype

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("cor_amd: the model must live on a GPU (model.to('cuda')); there is no CPU path")

    def _fingerprint(self, walk=True):
        """Cheap change detector for the packed-weight cache: every in-place update (optimizer step, p.data.copy_, a
        submodule's load_state_dict) bumps the tensor's _version; re-allocation (.to(), .half()) changes data_ptr.
        walk=False: over the tensor list of the last walk (the module walk is 1 ms of host time for 682 tensors, the hash 0.1 ms;
        a replayed graph checks every call and walks every 64th; _apply / load_state_dict drop the list)."""
        if walk or self._fp_tensors is None:
            self._fp_tensors = list(self.parameters()) + list(self.buffers())
        return hash(tuple((t._version, t.data_ptr()) for t in self._fp_tensors))

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
    def capture(self, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs, multimask_output=True,
                warmup=2, overlap_branches=True):
        """Capture forward() for THESE input shapes into a hipGraph and return a `GraphedForward`: calling it copies new
        inputs into the captured buffers and replays the ~600 kernel launches of a forward as ONE graph launch (the eager
        path pays one ctypes call + one launch per kernel: launch-bound at small batch). The example inputs are only read
        during warm-up and capture; the packed weights are captured by address (re-capture after changing parameters).
        overlap_branches: the support branch (SigLIP towers, adapter, fusion: ~200 small kernels that do not fill the chip) is
        captured on a second stream, i.e. as a parallel branch of the graph beside the SAM encoder (+3-4 % at batch 32)."""
        if self.training:
            raise RuntimeError("cor_amd implements the retrieval-time (inference) forward only: call model.eval() first")
        self._require_gpu()
        return GraphedForward(self, (query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs),
                              multimask_output, warmup, overlap_branches)

    @torch.no_grad()
    def capture_pipeline(self, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs, multimask_output=True,
                         depth=2, warmup=2, overlap_branches=True, stagger=False):
        """`depth` captured forwards with their OWN buffers, replayed round-robin on `depth` HIP streams (`ForwardPipeline`): the
        latency-bound end of forward i (support head, mask decoder: ~2.7 ms of small kernels at batch 32) runs beside the encoder
        GEMMs of forward i + 1 (+2.4 % throughput at batch 32 with depth 2; depth 3 measures lower).
        The gain depends on HOW the two forwards share the chip: with the runtime's default of 4 hardware queues (GPU_MAX_HW_QUEUES) the
        slots' streams alias onto shared queues and the forwards run staggered (one's encoder beside the other's tail); with 8 or 16
        queues they advance in lockstep and a step is 1.6-2.2 % slower (round 5, profiles/r05_pipeline_queues.jsonl; 8 % in round 4).
        `stagger=True` orders a slot's [encoder || support branch] graph behind the previous slot's by an event (two graphs per slot):
        measured equal at 4 queues and WORSE at 8 / 16 (the decoder chain then really runs beside the next encoder's persistent GEMMs and
        delays them), so it is off by default and a queue count other than 4 in the environment is reported with a warning."""
        q = os.environ.get("GPU_MAX_HW_QUEUES")
        if depth > 1 and q not in (None, "4"):
            warnings.warn(f"cor_amd.capture_pipeline: GPU_MAX_HW_QUEUES={q}: two forwards in flight were measured 1.6-2.2 % slower per step with 8 or 16 "
                          "hardware queues than with the runtime's default of 4 (profiles/r05_pipeline_queues.jsonl)", RuntimeWarning, stacklevel=2)
        if self.training:
            raise RuntimeError("cor_amd implements the retrieval-time (inference) forward only: call model.eval() first")
        self._require_gpu()
        return ForwardPipeline(self, (query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs),
                               multimask_output, depth, warmup, overlap_branches, stagger)

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


class GraphedForward:
    """A captured forward (CirSegModelWithQuerySupportFeat.capture). __call__ takes the same four inputs (same shapes / dtypes
    as at capture time; CPU tensors are copied over), replays the graph on the current stream and returns the three outputs.
    The outputs are the graph's OWN buffers: they are overwritten by the next replay - pass clone=True to get copies."""

    def __init__(self, model, inputs, multimask_output, warmup, overlap_branches=True, split=False):
        self.model, self.multimask_output, self.calls, self.split = model, multimask_output, 0, split
        dev = model.device
        self.T = model._resolve_dtype()
        self.fingerprint = model._fingerprint()
        with torch.cuda.device(dev):
            self.static_in = [t.detach().to(dev).clone() for t in inputs]
            W = model.packed(self.T)
            # the graph reads the packed weights BY ADDRESS: hold them, so that a cleared / rebuilt cache (a no-op model.to(dev), a
            # re-pack after an eager forward) can neither free the memory under the graph nor go unnoticed
            self.W = W
            args = (W, model.image_encoder.cfg, model.support_branch.siglip.cfg, model.support_branch.mask_pooling_name, self.T)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # warm-up off the capture: per-device kernel attributes, allocator pools
                for _ in range(max(1, warmup)):
                    engine.forward(*args, *self.static_in, multimask_output, overlap_branches=overlap_branches)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: API calls of OTHER host threads (e.g. a process group's watchdog) must not invalidate the capture
            if not split:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.static_out = engine.forward(*args, *self.static_in, multimask_output, overlap_branches=overlap_branches)
            else:
                # two graphs over ONE memory pool: [SAM encoder || support branch] and [mask decoder + output layout]; the pipeline
                # orders the next slot's first graph behind this slot's first graph (ForwardPipeline, stagger)
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self._mid = engine.forward_encode(*args, *self.static_in, overlap_branches=overlap_branches)
                self.graph_dec = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_dec, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    self.static_out = engine.forward_decode(W, model.image_encoder.cfg, self.T, *self._mid, multimask_output)

    def _check_and_copy(self, inputs):
        self.calls += 1
        if self.model._fingerprint(walk=self.calls % 64 == 1) != self.fingerprint:
            raise RuntimeError("cor_amd: the model's parameters changed (or moved) since capture(): capture again")
        if self.model._packed.get(self.T) is not self.W:
            # the cache was dropped (_apply / load_state_dict / invalidate_packed) or rebuilt: the next eager forward would pack NEW
            # tensors while this graph keeps reading the old ones - same parameters today, silently stale after the next update
            if self.model._packed.get(self.T) is None and self.model._packed.get("fp") in (None, self.fingerprint):
                self.model._packed = {"fp": self.fingerprint, **{k: v for k, v in self.model._packed.items() if k != "fp"}, self.T: self.W}
            else:
                raise RuntimeError("cor_amd: the model's packed weights were rebuilt since capture(): capture again")
        for dst, src in zip(self.static_in, inputs):
            if src.shape != dst.shape:
                raise ValueError(f"cor_amd: captured for input shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)

    @torch.no_grad()
    def __call__(self, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs, clone=False):
        with torch.cuda.device(self.model.device):
            self._check_and_copy((query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs))
            self.graph.replay()
            if self.split:
                self.graph_dec.replay()
        return tuple(t.clone() for t in self.static_out) if clone else self.static_out


class ForwardPipeline:
    """`depth` GraphedForwards (each with its own input / activation / output buffers) on `depth` streams, used round-robin
    (CirSegModelWithQuerySupportFeat.capture_pipeline). Work of consecutive submits overlaps on the GPU; everything that one
    submit enqueues (input copy, replay, the caller's `then`) is ordered on that slot's stream, so a slot's buffers are rewritten
    only after whatever `then` enqueued has read them. Results are identical to single forwards (tests/test_gpu_parity.py)."""

    def __init__(self, model, inputs, multimask_output=True, depth=2, warmup=2, overlap_branches=True, stagger=False):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.model, self.n, self.last, self.stagger, self.enc_done = model, 0, [None] * depth, bool(stagger) and depth > 1, None
        dev = model.device
        with torch.cuda.device(dev):
            self.slots = [(GraphedForward(model, inputs, multimask_output, warmup if i == 0 else 1, overlap_branches, split=self.stagger),
                           torch.cuda.Stream(device=dev)) for i in range(depth)]

    def next_inputs(self):
        """The input buffers of the slot the NEXT submit() uses: fill them in place on the CURRENT stream and call submit() without
        inputs to skip the copy. The current stream is made to wait (on the GPU, the host does not block) for that slot's previous
        submit, whose replay may still be reading them."""
        i = self.n % len(self.slots)
        if self.last[i] is not None:
            torch.cuda.current_stream(self.model.device).wait_event(self.last[i])
        return self.slots[i][0].static_in

    @torch.no_grad()
    def submit(self, inputs=None, then=None):
        """Enqueue one forward on the next slot. inputs: the four tensors (copied into the slot's buffers) or None (the slot's
        buffers already hold them). then(outputs): called with the slot's stream current, to enqueue the consumer of the outputs
        (a search, a copy to the host); the outputs are that slot's buffers and stay valid until the slot's next submit.
        -> (outputs, then's return value, event recorded on the slot's stream after both)."""
        i = self.n % len(self.slots)
        g, st = self.slots[i]
        self.n += 1
        dev = self.model.device
        with torch.cuda.device(dev):
            st.wait_stream(torch.cuda.current_stream(dev))     # inputs written by the caller's stream are complete before the replay
            with torch.cuda.stream(st):
                if not self.stagger:
                    out = g(*(inputs if inputs is not None else g.static_in))
                else:
                    # explicit stagger: this slot's [encoder || support branch] graph starts when the previous submit's has ended (two
                    # encoders sharing the chip alternate kernel by kernel and both finish late); its decoder then runs beside the NEXT
                    # submit's encoder. No dependence on how the runtime maps streams to hardware queues.
                    g._check_and_copy(inputs if inputs is not None else g.static_in)
                    if self.enc_done is not None:
                        st.wait_event(self.enc_done)
                    g.graph.replay()
                    self.enc_done = torch.cuda.Event()
                    self.enc_done.record(st)
                    g.graph_dec.replay()
                    out = g.static_out
                res = then(out) if then is not None else None
                ev = torch.cuda.Event()
                ev.record(st)
        self.last[i] = ev
        return out, res, ev

