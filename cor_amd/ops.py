"""Torch-tensor front end of the C ABI (include/cor_amd.h).

torch is used here only for device memory and the current HIP stream; every function enqueues one hand-written
gfx950 kernel through libcor_amd.so. Operand shapes are validated on the host BEFORE the launch (a faulting
kernel can reset the GPU). There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import torch

from . import _native as nat
from ._native import ACT_NONE, ACT_GELU_ERF, ACT_RELU, ACT_SIGMOID, ACT_GELU_TANH, F32, BF16, F16  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

# bench.py sets this to a list to time every GEMM launch with HIP events recorded on the launch stream:
# entries are (start_event, end_event, algorithmic_flops, ab_dtype, algorithmic_bytes). None = no instrumentation.
GEMM_PROFILE = None


def _lib():
    return nat.load()


def _dt(t: torch.Tensor) -> int:
    return _DT[t.dtype]


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _dev(*ts):
    """Host-side guard run BEFORE every launch: the kernels take raw pointers and are enqueued on the CURRENT device's
    current stream, so (1) no CPU tensor, (2) all operands on ONE GPU, (3) that GPU is the current HIP device (otherwise
    the launch would dereference another device's memory: a GPU fault). Callers holding a model on another device wrap
    the call in `with torch.cuda.device(model.device)` (CirSegModelWithQuerySupportFeat.forward does)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cor_amd ops run on the GPU only (no CPU fallback): got a CPU tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"cor_amd ops: operands live on different devices ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"cor_amd ops: operands live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"wrap the call in `with torch.cuda.device({dev.index})`")
    return dev


def _rows(t: torch.Tensor):
    """(rows, cols, ld) of a 2-D tensor whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, f"expected 2-D row-major view, got {tuple(t.shape)} / {t.stride()}"
    return t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _f32vec(t, n, name):
    if t is None:
        return
    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, f"{name}: need contiguous fp32[{n}]"


def gemm(a, w, out_dtype=None, bias=None, act=ACT_NONE, col_scale=None, residual=None, res_row_mod=0, out=None, cfg=0, reverse=False):
    """out[M,N] = residual + col_scale * act(a[M,K] @ w[N,K]^T + bias). cfg: per-call kernel choice (0 = automatic).
    reverse: walk the tiles from the last row panel to the first (same result; see ORDER_REVERSE)."""
    _dev(a, w, bias, col_scale, residual, out)
    M, K, lda = _rows(a)
    N, K2, ldw = _rows(w)
    assert K == K2 and a.dtype == w.dtype, (a.shape, w.shape, a.dtype, w.dtype)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
    Mo, No, ldc = _rows(out)
    assert (Mo, No) == (M, N)
    _f32vec(bias, N, "bias")
    _f32vec(col_scale, N, "col_scale")
    ldr = 0
    if residual is not None:
        Mr, Nr, ldr = _rows(residual)
        assert residual.dtype == torch.float32 and Nr == N and Mr == (res_row_mod if res_row_mod > 0 else M)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    nat.check(_lib().cor_gemm(a.data_ptr(), lda, w.data_ptr(), ldw, _dt(a), out.data_ptr(), ldc, _dt(out), M, N, K,
                              _p(bias), act, _p(col_scale), _p(residual), ldr, res_row_mod, int(cfg) | (nat.ORDER_REVERSE if reverse else 0), _s()), "cor_gemm")
    if prof is not None:
        e1.record()
        nbytes = (M * K + N * K) * a.element_size() + M * N * out.element_size() + (M * N * 4 if residual is not None else 0)
        prof.append((e0, e1, 2.0 * M * N * K, a.dtype, float(nbytes)))
    return out


def layernorm(x, w, b, eps, out_dtype=None, act=ACT_NONE, out=None, reverse=False):
    _dev(x, w, b, out)
    assert x.is_contiguous() and x.dim() == 2
    rows, C = x.shape
    _f32vec(w, C, "ln.weight")
    _f32vec(b, C, "ln.bias")
    if out is None:
        out = torch.empty((rows, C), dtype=out_dtype or x.dtype, device=x.device)
    assert out.is_contiguous() and out.shape == x.shape
    nat.check(_lib().cor_layernorm(x.data_ptr(), _dt(x), out.data_ptr(), _dt(out), w.data_ptr(), b.data_ptr(), rows, C,
                                   float(eps), act | (nat.ORDER_REVERSE if reverse else 0), _s()), "cor_layernorm")
    return out


def attention(q, k, v, B, H, Tq, Tk, hd, scale, out_dtype=None):
    """q [B*Tq, >=H*hd] / k,v [B*Tk, ...] row-major 2-D views (e.g. column slices of a fused qkv activation)."""
    _dev(q, k, v)
    assert q.dtype == k.dtype == v.dtype
    for t, T in ((q, Tq), (k, Tk), (v, Tk)):
        assert t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == B * T and t.shape[1] == H * hd, (t.shape, B, T, H, hd)
    out = torch.empty((B * Tq, H * hd), dtype=out_dtype or q.dtype, device=q.device)
    nat.check(_lib().cor_attention(q.data_ptr(), Tq * q.stride(0), q.stride(0), k.data_ptr(), Tk * k.stride(0), k.stride(0),
                                   v.data_ptr(), Tk * v.stride(0), v.stride(0), _dt(q), out.data_ptr(), Tq * H * hd, H * hd,
                                   _dt(out), B, H, Tq, Tk, hd, float(scale), _s()), "cor_attention")
    return out


def sam_attention(qkv, pad_row, rel_h, rel_w, B, H, grid, window, out_dtype=None, variant=0, q_prescale=1.0, reverse=False):
    _dev(qkv, pad_row, rel_h, rel_w)
    hd = rel_h.shape[1]
    d = H * hd
    assert qkv.is_contiguous() and qkv.shape == (B * grid * grid, 3 * d), qkv.shape
    S = window if window > 0 else grid
    for r in (rel_h, rel_w):
        assert r.dtype == torch.float32 and r.is_contiguous() and r.shape == (2 * S - 1, hd), r.shape
    if window > 0:
        assert pad_row is not None and pad_row.dtype == qkv.dtype and pad_row.is_contiguous() and pad_row.numel() == 3 * d
    out = torch.empty((B * grid * grid, d), dtype=out_dtype or qkv.dtype, device=qkv.device)
    nat.check(_lib().cor_sam_attention(qkv.data_ptr(), _dt(qkv), out.data_ptr(), _dt(out), _p(pad_row), rel_h.data_ptr(),
                                       rel_w.data_ptr(), B, H, hd, grid, window, float(q_prescale), int(variant) | (nat.ORDER_REVERSE if reverse else 0), _s()), "cor_sam_attention")
    return out


def patchify(img, p, Kpad, out_dtype):
    _dev(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4
    B, C, H, W = img.shape
    out = torch.empty((B * (H // p) * (W // p), Kpad), dtype=out_dtype, device=img.device)
    nat.check(_lib().cor_patchify(img.data_ptr(), out.data_ptr(), _dt(out), B, C, H, W, p, Kpad, _s()), "cor_patchify")
    return out


def im2col3x3(x, B, H, W):
    _dev(x)
    C = x.shape[-1]
    assert x.is_contiguous() and x.numel() == B * H * W * C
    out = torch.empty((B * H * W, 9 * C), dtype=x.dtype, device=x.device)
    nat.check(_lib().cor_im2col3x3(x.data_ptr(), _dt(x), out.data_ptr(), B, H, W, C, _s()), "cor_im2col3x3")
    return out


def add(a, b, out_dtype=None, out=None):
    """a + b, b broadcast periodically over the flattened a (b.numel() must divide a.numel())."""
    _dev(a, b, out)
    assert a.is_contiguous() and b.is_contiguous() and a.numel() % b.numel() == 0
    if out is None:
        out = torch.empty(a.shape, dtype=out_dtype or a.dtype, device=a.device)
    assert out.is_contiguous() and out.numel() == a.numel()
    nat.check(_lib().cor_add(a.data_ptr(), _dt(a), b.data_ptr(), _dt(b), out.data_ptr(), _dt(out), a.numel(), b.numel(), _s()),
              "cor_add")
    return out


def copy_rows(src_ptr_tensor, ld_in, rows, C, out, ld_out=None, src_offset=0):
    """out[r, :C] = src[src_offset + r*ld_in : ... + C]; casts between fp32/bf16. ld_in == 0 broadcasts one row."""
    _dev(src_ptr_tensor, out)
    src = src_ptr_tensor
    assert src.is_contiguous()
    need = src_offset + (rows - 1) * ld_in + C
    assert need <= src.numel(), (need, src.numel())
    if ld_out is None:
        ld_out = out.stride(0) if out.dim() == 2 else C
    assert out.stride(-1) == 1
    nat.check(_lib().cor_copy_rows(src.data_ptr() + src_offset * src.element_size(), ld_in, _dt(src), out.data_ptr(), ld_out,
                                   _dt(out), rows, C, _s()), "cor_copy_rows")
    return out


def cast(x, dtype):
    """Contiguous dtype conversion through cor_copy_rows (no-op when already `dtype`)."""
    if x.dtype == dtype:
        return x
    assert x.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    return copy_rows(x, C, rows, C, out, ld_out=C)


def tokens_to_nchw(x, B, HW, C):
    _dev(x)
    assert x.is_contiguous() and x.numel() == B * HW * C
    out = torch.empty((B, C, HW), dtype=torch.float32, device=x.device)
    nat.check(_lib().cor_tokens_to_nchw(x.data_ptr(), _dt(x), out.data_ptr(), B, HW, C, _s()), "cor_tokens_to_nchw")
    return out


def nchw_to_tokens(x, out_dtype):
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    out = torch.empty((B * HW, C), dtype=out_dtype, device=x.device)
    nat.check(_lib().cor_nchw_to_tokens(x.data_ptr(), out.data_ptr(), _dt(out), B, HW, C, _s()), "cor_nchw_to_tokens")
    return out


def l2norm_rows(x, eps=1e-12, out_dtype=None):
    _dev(x)
    assert x.is_contiguous() and x.dim() == 2
    out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    nat.check(_lib().cor_l2norm_rows(x.data_ptr(), _dt(x), out.data_ptr(), _dt(out), x.shape[0], x.shape[1], float(eps), _s()),
              "cor_l2norm_rows")
    return out


def embed_tokens(ids, table, pos):
    _dev(ids, table, pos)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2
    N, ctx = ids.shape
    vocab, D = table.shape
    assert table.dtype == torch.float32 and table.is_contiguous() and pos.dtype == torch.float32 and pos.is_contiguous()
    assert pos.shape[0] >= ctx and pos.shape[1] == D
    out = torch.empty((N * ctx, D), dtype=torch.float32, device=ids.device)
    nat.check(_lib().cor_embed_tokens(ids.data_ptr(), table.data_ptr(), pos.data_ptr(), out.data_ptr(), N * ctx, ctx, D, vocab, _s()),
              "cor_embed_tokens")
    return out


def bilinear(x, OH, OW, clamp01=False):
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, C, H, W = x.shape
    out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    nat.check(_lib().cor_bilinear(x.data_ptr(), out.data_ptr(), B * C, H, W, OH, OW, int(clamp01), _s()), "cor_bilinear")
    return out


def conv3x3s2_small(x, channels_last, w, bias, B, Cin, H, W):
    _dev(x, w, bias)
    Cout = w.shape[0]
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == B * Cin * H * W
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape == (Cout, Cin, 3, 3)
    _f32vec(bias, Cout, "conv.bias")
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    nat.check(_lib().cor_conv3x3s2_small(x.data_ptr(), int(channels_last), w.data_ptr(), _p(bias), out.data_ptr(), B, Cin, Cout, H, W,
                                         _s()), "cor_conv3x3s2_small")
    return out


def dwconv7x7(x, w_t, bias, B, H, W, out_dtype=torch.float32):
    _dev(x, w_t, bias)
    C = x.shape[-1]
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == B * H * W * C
    assert w_t.dtype == torch.float32 and w_t.is_contiguous() and w_t.shape == (49, C)
    _f32vec(bias, C, "dwconv.bias")
    out = torch.empty((B * H * W, C), dtype=out_dtype, device=x.device)
    nat.check(_lib().cor_dwconv7x7(x.data_ptr(), w_t.data_ptr(), bias.data_ptr(), out.data_ptr(), _dt(out), B, H, W, C, _s()),
              "cor_dwconv7x7")
    return out


def adapter_pool(maps, feat, B, P, M, D):
    _dev(maps, feat)
    assert maps.dtype == torch.float32 and maps.is_contiguous() and maps.numel() == B * P * M
    assert feat.dtype == torch.float32 and feat.is_contiguous() and feat.numel() == B * P * D
    out = torch.empty((B, D), dtype=torch.float32, device=maps.device)
    nat.check(_lib().cor_adapter_pool(maps.data_ptr(), feat.data_ptr(), out.data_ptr(), B, P, M, D, _s()), "cor_adapter_pool")
    return out


def masked_pool(feat, mask, B, P, D, feat_nchw=False, clamp01=False, l2norm=False):
    _dev(feat, mask)
    assert feat.dtype == torch.float32 and feat.is_contiguous() and feat.numel() == B * P * D
    assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.numel() == B * P
    out = torch.empty((B, D), dtype=torch.float32, device=feat.device)
    nat.check(_lib().cor_masked_pool(feat.data_ptr(), int(feat_nchw), mask.data_ptr(), out.data_ptr(), B, P, D, int(clamp01),
                                     int(l2norm), _s()), "cor_masked_pool")
    return out


def fuse_gate(img, txt, aI, aT):
    _dev(img, txt, aI, aT)
    N, D = img.shape
    for t in (img, txt, aI, aT):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (N, D)
    cat = torch.empty((N, 2 * D), dtype=torch.float32, device=img.device)
    nat.check(_lib().cor_fuse_gate(img.data_ptr(), txt.data_ptr(), aI.data_ptr(), aT.data_ptr(), cat.data_ptr(), N, D, _s()),
              "cor_fuse_gate")
    return cat


def fuse_mix(cat, dyn):
    _dev(cat, dyn)
    N, D2 = cat.shape
    assert cat.dtype == torch.float32 and cat.is_contiguous() and dyn.dtype == torch.float32 and dyn.is_contiguous() and dyn.numel() == N
    out = torch.empty((N, D2 // 2), dtype=torch.float32, device=cat.device)
    nat.check(_lib().cor_fuse_mix(cat.data_ptr(), dyn.data_ptr(), out.data_ptr(), N, D2 // 2, _s()), "cor_fuse_mix")
    return out


def dense_pe(gauss, size):
    _dev(gauss)
    assert gauss.dtype == torch.float32 and gauss.is_contiguous() and gauss.dim() == 2 and gauss.shape[0] == 2
    F = gauss.shape[1]
    out = torch.empty((size * size, 2 * F), dtype=torch.float32, device=gauss.device)
    nat.check(_lib().cor_dense_pe(gauss.data_ptr(), out.data_ptr(), size, F, _s()), "cor_dense_pe")
    return out


def upscale_shuffle(y, B, H, W, Cout, bias=None, ln_w=None, ln_b=None, eps=1e-6, act=ACT_NONE, out_dtype=None):
    _dev(y, bias, ln_w, ln_b)
    assert y.is_contiguous() and y.shape == (B * H * W, 4 * Cout)
    _f32vec(bias, Cout, "bias")
    _f32vec(ln_w, Cout, "ln_w")
    _f32vec(ln_b, Cout, "ln_b")
    out = torch.empty((B * 4 * H * W, Cout), dtype=out_dtype or y.dtype, device=y.device)
    nat.check(_lib().cor_upscale_shuffle(y.data_ptr(), _dt(y), _p(bias), _p(ln_w), _p(ln_b), float(eps), act, out.data_ptr(), _dt(out),
                                         B, H, W, Cout, _s()), "cor_upscale_shuffle")
    return out


def upscale_hyper(x, w, bias, hyper, B, H, W, Kmask):
    """x [B*H*W, 64] -> masks [B, Kmask, 2H, 2W]; hyper [B, Kmask, 32] fp32 (may be a strided view over dim 0)."""
    _dev(x, w, bias, hyper)
    assert x.is_contiguous() and x.shape == (B * H * W, 64)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape == (64, 32, 2, 2)
    _f32vec(bias, 32, "bias")
    assert hyper.dtype == torch.float32 and hyper.shape == (B, Kmask, 32) and hyper.stride(2) == 1 and hyper.stride(1) == 32
    masks = torch.empty((B, Kmask, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    nat.check(_lib().cor_upscale_hyper(x.data_ptr(), _dt(x), w.data_ptr(), bias.data_ptr(), hyper.data_ptr(),
                                       hyper.stride(0) if B > 1 else Kmask * 32, masks.data_ptr(), B, H, W, 64, 32, Kmask, _s()),
              "cor_upscale_hyper")
    return masks


def iou_select(iou, hyper, k_off, Ksel):
    _dev(iou, hyper)
    B, Kall = iou.shape
    C = hyper.shape[-1]
    assert iou.dtype == torch.float32 and iou.is_contiguous() and hyper.dtype == torch.float32 and hyper.is_contiguous()
    assert hyper.shape == (B, Kall, C)
    best = torch.empty((B,), dtype=torch.int64, device=iou.device)
    sel = torch.empty((B, 1, C), dtype=torch.float32, device=iou.device)
    nat.check(_lib().cor_iou_select(iou.data_ptr(), hyper.data_ptr(), B, Kall, k_off, Ksel, C, best.data_ptr(), sel.data_ptr(), _s()),
              "cor_iou_select")
    return best, sel


def similarity_topk(Q, G, k, g_offset=0, flags=0):
    """Top-k gallery rows per query by dot product; (score desc, index asc). Q fp32 [Bq,C]; G [Ng,C] fp32/bf16/fp16.
    fp32 galleries and 16-bit galleries with C = 256: scores and indices are bit-identical to the CPU fmaf-chain oracle
    (oracle/c/sim_chain.c). A candidate overflow (pathological score distributions) is repaired ON THE DEVICE by the gated
    list kernels: no host synchronisation here. flags: nat.TOPK_FORCE_LISTS | nat.TOPK_NO_FALLBACK (tests)."""
    _dev(Q, G)
    assert Q.dtype == torch.float32 and Q.is_contiguous() and G.is_contiguous() and Q.dim() == 2 and G.dim() == 2
    Bq, Cq = Q.shape
    Ng, Cg = G.shape
    assert Cq == Cg
    lib = _lib()
    nbytes = lib.cor_topk_workspace_bytes(Bq, Ng, k)
    if nbytes < 0:
        nat.check(int(nbytes), "cor_topk_workspace_bytes")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=Q.device)          # torch's allocator returns >= 256-B aligned blocks
    scores = torch.empty((Bq, k), dtype=torch.float32, device=Q.device)
    idx = torch.empty((Bq, k), dtype=torch.int64, device=Q.device)
    nat.check(lib.cor_similarity_topk(Q.data_ptr(), G.data_ptr(), _dt(G), Bq, Ng, Cq, k, int(g_offset), scores.data_ptr(),
                                      idx.data_ptr(), ws.data_ptr(), int(flags), _s()), "cor_similarity_topk")
    return scores, idx


def decoder_heads(hs, w01, b01, w2, b2):
    """The mask decoder's five output MLPs in one launch (cor_decoder_heads). hs [B*6,256] in the weights' dtype ->
    (hyper f32 [B,4,32], iou f32 [B,4])."""
    _dev(hs, w01, b01, w2, b2)
    assert hs.is_contiguous() and hs.dim() == 2 and hs.shape[1] == 256 and hs.shape[0] % 6 == 0 and hs.dtype == w01.dtype == w2.dtype
    assert tuple(w01.shape) == (5, 2, 256, 256) and tuple(b01.shape) == (5, 2, 256) and tuple(w2.shape) == (132, 256) and tuple(b2.shape) == (132,)
    assert w01.is_contiguous() and w2.is_contiguous() and b01.is_contiguous() and b2.is_contiguous() and b01.dtype == b2.dtype == torch.float32
    B = hs.shape[0] // 6
    hyper = torch.empty((B, 4, 32), dtype=torch.float32, device=hs.device)
    iou = torch.empty((B, 4), dtype=torch.float32, device=hs.device)
    nat.check(_lib().cor_decoder_heads(hs.data_ptr(), w01.data_ptr(), b01.data_ptr(), w2.data_ptr(), b2.data_ptr(), _dt(hs), hyper.data_ptr(),
                                       iou.data_ptr(), B, _s()), "cor_decoder_heads")
    return hyper, iou


def mask_prob_minmax(logits):
    """sigmoid + per-sample min-max normalisation of mask logits [B,1,H,W] (utils/vailder.py:426-430)."""
    _dev(logits)
    x = logits.to(torch.float32).contiguous()
    B = x.shape[0]
    out = torch.empty_like(x)
    nat.check(_lib().cor_mask_prob_minmax(x.data_ptr(), out.data_ptr(), B, x.numel() // B, _s()), "cor_mask_prob_minmax")
    return out


def resize_binarize(prob, OH, OW, threshold=0.5):
    """[B,1,H,W] probabilities -> uint8 {0,255} [B,OH,OW] (cv2.INTER_LINEAR semantics, utils/vailder.py:459-473)."""
    _dev(prob)
    assert prob.dtype == torch.float32 and prob.is_contiguous()
    B, H, W = prob.shape[0], prob.shape[-2], prob.shape[-1]
    assert prob.numel() == B * H * W
    out = torch.empty((B, OH, OW), dtype=torch.uint8, device=prob.device)
    nat.check(_lib().cor_resize_binarize(prob.data_ptr(), out.data_ptr(), B, H, W, OH, OW, float(threshold), _s()), "cor_resize_binarize")
    return out


def resize_gray(prob, OH, OW):
    """[B,1,H,W] probabilities -> uint8 [B,OH,OW] = (resized * 255) truncated (cv2.INTER_LINEAR semantics, utils/vailder.py:615-621)."""
    _dev(prob)
    assert prob.dtype == torch.float32 and prob.is_contiguous()
    B, H, W = prob.shape[0], prob.shape[-2], prob.shape[-1]
    assert prob.numel() == B * H * W
    out = torch.empty((B, OH, OW), dtype=torch.uint8, device=prob.device)
    nat.check(_lib().cor_resize_gray(prob.data_ptr(), out.data_ptr(), B, H, W, OH, OW, _s()), "cor_resize_gray")
    return out


def mask_metrics(pred, gt, smooth=1e-5):
    """per-sample [dice, mae, iou, mdice, miou] (utils/trainer_v3_g.py:381-443). pred, gt: [B,...] fp32 of equal shape."""
    _dev(pred, gt)
    assert pred.shape == gt.shape, f"Shape mismatch: pred {tuple(pred.shape)} vs gt {tuple(gt.shape)}"
    p, g = pred.to(torch.float32).contiguous(), gt.to(torch.float32).contiguous()
    B = p.shape[0]
    out = torch.empty((B, 5), dtype=torch.float32, device=p.device)
    nat.check(_lib().cor_mask_metrics(p.data_ptr(), g.data_ptr(), out.data_ptr(), B, p.numel() // B, float(smooth), _s()), "cor_mask_metrics")
    return out
