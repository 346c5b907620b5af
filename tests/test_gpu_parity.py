"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel / stage / the whole forward, called through the
C ABI (cor_amd.ops -> libcor_amd.so), against (a) golden vectors produced by the REFERENCE's own modules and
(b) the CPU oracle on seeded inputs. Tolerances are stated per test: fp32 (exact-MFMA) mode is held to the
north-star 1e-3; bf16 mode has its own, looser, documented budget.

Every comparison also lands in gpurun_out/parity_report.jsonl (max abs / rel error) for DESIGN.md."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import config as ocfg, sam as osam, siglip as osig, support as osup, retrieval as oret, model as omodel
from tests.golden_util import load, make_inputs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")
DEV = "cuda:0"
F32, BF16 = torch.float32, torch.bfloat16


def _ops():
    from cor_amd import ops, engine
    return ops, engine


def report(name, got, ref, rtol, atol):
    got = got.detach().float().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(np.asarray(got)).float()
    ref = ref.detach().float().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    rec = dict(name=name, max_abs=float(err.max()), max_ref=float(ref.abs().max()), mean_abs=float(err.mean()),
               rel_l2=float((got - ref).norm() / (ref.norm() + 1e-30)), finite=bool(torch.isfinite(got).all()), rtol=rtol, atol=atol)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    bad = err > (atol + rtol * ref.abs())
    assert rec["finite"], f"{name}: non-finite output"
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance; {rec}"
    return rec


def dev(sd):
    return {k: v.to(DEV) for k, v in sd.items()}


def packer(sd, T):
    _, engine = _ops()
    return engine._Packer(dev(sd), T)


# ======================================================================================================
# library / ABI
# ======================================================================================================
def test_native_library_is_the_one_in_tree():
    from cor_amd import _native
    lib = _native.load()
    assert lib.cor_version() >= 1
    assert os.path.samefile(_native.LIB_PATH, os.path.join(ROOT, "cor_amd", "csrc", "libcor_amd.so"))


def test_cpu_tensor_is_refused():
    ops, _ = _ops()
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 16), torch.zeros(4, 16))


# ======================================================================================================
# GEMM
# ======================================================================================================
GEMM_SHAPES = [(128, 128, 128), (300, 200, 256), (1000, 768, 3072), (7, 1, 768), (64, 8, 256), (4096, 2304, 768),
               (33, 130, 588), (2, 512, 16), (576, 256, 1024)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("T", [F32, BF16])
def test_gemm_plain(M, N, K, T):
    ops, _ = _ops()
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(T)
    w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32)).to(T)
    ref = a.float().double() @ w.float().double().T
    out = ops.gemm(a.to(DEV), w.to(DEV), out_dtype=F32)
    # inputs are identical (already rounded to T); products exact in fp32; only the summation order differs
    report(f"gemm_plain_{M}x{N}x{K}_{T}", out, ref.float(), rtol=1e-5, atol=2e-5 * math.sqrt(K) * 4)


@pytest.mark.parametrize("T,TO", [(F32, F32), (BF16, BF16), (BF16, F32)])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_epilogue(T, TO, act):
    ops, _ = _ops()
    M, N, K = 260, 200, 192
    rng = np.random.default_rng(act + 11)
    a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(T)
    w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).to(T)
    bias = torch.from_numpy(rng.standard_normal(N, dtype=np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((52, N), dtype=np.float32))      # periodic residual (M = 5 * 52)
    z = a.float() @ w.float().T + bias
    actf = [lambda v: v, lambda v: torch.nn.functional.gelu(v), torch.relu, torch.sigmoid,
            lambda v: torch.nn.functional.gelu(v, approximate="tanh")][act]
    ref = actf(z) * scale + res.repeat(5, 1)
    out = ops.gemm(a.to(DEV), w.to(DEV), out_dtype=TO, bias=bias.to(DEV), act=act, col_scale=scale.to(DEV),
                   residual=res.to(DEV), res_row_mod=52)
    tol = dict(rtol=1e-4, atol=1e-4) if TO == F32 else dict(rtol=1e-2, atol=1e-2)
    report(f"gemm_epi_act{act}_{T}_{TO}", out, ref, **tol)


def test_gemm_inplace_residual_and_strided_views():
    ops, _ = _ops()
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((500, 256), dtype=np.float32)).to(DEV)
    a = torch.from_numpy(rng.standard_normal((500, 1536), dtype=np.float32)).to(DEV)
    w = torch.from_numpy(rng.standard_normal((256, 512), dtype=np.float32) / 16).to(DEV)
    ref = x.cpu() + a.cpu()[:, 512:1024] @ w.cpu().T
    ops.gemm(a[:, 512:1024], w, out_dtype=F32, residual=x, out=x)          # A is a column slice; C aliases the residual
    report("gemm_inplace_strided", x, ref, rtol=1e-4, atol=1e-4)


# The persistent 256x256 ping-pong kernel (gemm_pp) is chosen for bf16 operands once the output has >= 200 tiles of 256x256:
# none of the shapes above reach it, so it gets its own cases. K = 64 / 128 / 192 run one / two / three K-tiles (prologue-only,
# first in-loop wait, steady state); M and N are ragged against 256; every case is also compared bit for bit with the 128x128
# kernel (same accumulation order per k-step).
@pytest.mark.parametrize("M,N,K", [(16645, 1032, 64), (16645, 1032, 128), (16645, 1032, 192), (12800, 1288, 576), (70000, 776, 320)])
@pytest.mark.parametrize("mode", ["bf16", "bf16_gelu", "f32", "f32_res", "f32_res_inplace", "bf16_res_periodic", "bf16_gelu_tanh"])
def test_gemm_pingpong(M, N, K, mode):
    ops, _ = _ops()
    from cor_amd import _native as nat
    lib = nat.load()
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 200
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + 7 * K)
    a = torch.randn((M, K), generator=g, device=DEV).to(BF16)
    w = (torch.randn((N, K), generator=g, device=DEV) / math.sqrt(K)).to(BF16)
    bias = torch.randn((N,), generator=g, device=DEV)
    TO = BF16 if mode.startswith("bf16") else F32
    act = 1 if mode == "bf16_gelu" else (4 if mode == "bf16_gelu_tanh" else 0)
    period = 325 if mode == "bf16_res_periodic" else 0
    res = None
    if "res" in mode:
        res = torch.randn((period or M, N), generator=g, device=DEV)
    def run(cfg):                                                  # the kernel choice is a per-call argument (no global state)
        if mode == "f32_res_inplace":
            x = res.clone()
            ops.gemm(a, w, out_dtype=F32, bias=bias, residual=x, out=x, cfg=cfg)
            return x
        return ops.gemm(a, w, out_dtype=TO, bias=bias, act=act, residual=res, res_row_mod=period, cfg=cfg)
    out = run(0)                                                   # auto: must pick the ping-pong kernel
    out128 = run(2)
    assert torch.equal(out, out128), f"ping-pong vs 128x128 kernel differ: {(out.float() - out128.float()).abs().max().item()}"
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,))]).to(DEV)   # both edges + a sample
    z = a[rows].float() @ w.float().T + bias
    if act == 1: z = torch.nn.functional.gelu(z)
    if act == 4: z = torch.nn.functional.gelu(z, approximate="tanh")
    if res is not None: z = z + (res[rows % period] if period else res[rows])
    tol = dict(rtol=1e-4, atol=2e-4) if TO == F32 else dict(rtol=1e-2, atol=2e-2)
    report(f"gemm_pp_{M}x{N}x{K}_{mode}", out[rows], z, **tol)


def test_gemm_pingpong_guard_rows_and_fallbacks():
    """The buffer-store epilogue must not touch rows around a C view; column-scaled / K % 64 != 0 GEMMs fall back to the 128x128 kernels."""
    ops, _ = _ops()
    M, N, K = 16500, 1040, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn((M, K), generator=g, device=DEV).to(BF16)
    w = (torch.randn((N, K), generator=g, device=DEV) / 16).to(BF16)
    for TO in (BF16, F32):
        guard = torch.full((M + 2, N), 7.0, device=DEV, dtype=TO)
        out = ops.gemm(a, w, out_dtype=TO, out=guard[1:M + 1])
        assert bool((guard[0] == 7).all()) and bool((guard[M + 1] == 7).all())
        report(f"gemm_pp_guard_{TO}", out[:256], a[:256].float() @ w.float().T, **(dict(rtol=1e-4, atol=2e-4) if TO == F32 else dict(rtol=1e-2, atol=2e-2)))
    scale = torch.rand((N,), generator=g, device=DEV) + 0.5
    out = ops.gemm(a, w, out_dtype=F32, col_scale=scale)
    report("gemm_pp_fallback_colscale", out[:256], (a[:256].float() @ w.float().T) * scale, rtol=1e-4, atol=2e-4)
    a2, w2 = a[:, :200].contiguous(), w[:, :200].contiguous()      # K = 200: not a multiple of 64
    report("gemm_pp_fallback_ktail", ops.gemm(a2, w2, out_dtype=F32)[:256], a2[:256].float() @ w2.float().T, rtol=1e-4, atol=2e-4)


def test_configs3_shapes_kernel_by_kernel():
    """BASELINE configs[3] (SAM-L + SigLIP-L/16, batch 64) launches its kernels at M = 262 144 rows: the first full-size run of
    that configuration in round 2 ended in a process abort at the first synchronisation after the forward (gpurun_out/pytest_a.log,
    08:24; cause not recoverable from the log). This test runs every large launch of that forward ALONE
    at exactly those shapes and checks sampled rows against torch / fp64 references, with a synchronisation after each, so that a
    device-side failure is pinned to ONE launch: byte offsets there reach 2^31 (lin1 output, lin2 A operand), 32-bit offset
    arithmetic is host-checked to stay below 2^32."""
    ops, _ = _ops()
    M, d, H = 64 * 4096, 1024, 16
    g = torch.Generator(device=DEV).manual_seed(3)
    rows = torch.cat([torch.arange(0, 256), torch.arange(M - 256, M), torch.randint(0, M, (256,))]).to(DEV)
    x = torch.randn((M, d), generator=g, device=DEV)
    lnw, lnb = torch.rand((d,), generator=g, device=DEV) + 0.5, torch.randn((d,), generator=g, device=DEV) * 0.1
    h = ops.layernorm(x, lnw, lnb, 1e-6, out_dtype=BF16, reverse=True); torch.cuda.synchronize()
    report("cfg3_layernorm_262144x1024", h[rows], torch.nn.functional.layer_norm(x[rows], (d,), lnw, lnb, 1e-6), rtol=1e-2, atol=2e-2)
    def lin(n, k):
        return (torch.randn((n, k), generator=g, device=DEV) / math.sqrt(k)).to(BF16), torch.randn((n,), generator=g, device=DEV) * 0.1
    for name, n, k, kind in (("qkv", 3 * d, d, "bf16"), ("proj_res", d, d, "res"), ("lin1_gelu", 4 * d, d, "gelu"), ("lin2_res", d, 4 * d, "res")):
        w, b = lin(n, k)
        a = h if k == d else torch.randn((M, k), generator=g, device=DEV).to(BF16)
        z = a[rows].float() @ w.float().T + b
        if kind == "res":
            xr = x.clone()
            ops.gemm(a, w, out_dtype=F32, bias=b, residual=xr, out=xr); torch.cuda.synchronize()
            report(f"cfg3_gemm_{name}_262144x{n}x{k}", xr[rows], z + x[rows], rtol=1e-4, atol=2e-3)
            del xr
        else:
            y = ops.gemm(a, w, out_dtype=BF16, bias=b, act=ops.ACT_GELU_ERF if kind == "gelu" else ops.ACT_NONE, reverse=(kind == "gelu")); torch.cuda.synchronize()
            report(f"cfg3_gemm_{name}_262144x{n}x{k}", y[rows], torch.nn.functional.gelu(z) if kind == "gelu" else z, rtol=1e-2, atol=3e-2)
            if name == "qkv":
                qkv = y
        del a, w
    pad = torch.randn((3 * d,), generator=g, device=DEV).to(BF16)
    for win, S in ((14, 14), (0, 64)):
        rh, rw = torch.randn((2 * S - 1, 64), generator=g, device=DEV) * 0.3, torch.randn((2 * S - 1, 64), generator=g, device=DEV) * 0.3
        o = ops.sam_attention(qkv, pad, rh, rw, 64, H, 64, win, reverse=True); torch.cuda.synchronize()
        o1 = ops.sam_attention(qkv, pad, rh, rw, 64, H, 64, win, variant=1); torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        report(f"cfg3_sam_attention_window{win}_B64_H16", o, o1, rtol=2e-2, atol=2e-2 * float(o1.float().abs().max()))
        if win == 0:                                   # fp64 reference for the LAST image and head (largest offsets)
            xq = qkv.view(64, 4096, 3, H, 64)
            q, k, v = (xq[63, :, j, H - 1].double().cpu() for j in range(3))
            bias = osam.rel_pos_bias(q[None], rh.to(BF16).double().cpu(), rw.to(BF16).double().cpu(), 64)[0]
            ref = torch.softmax((q * 0.125) @ k.T + bias, dim=-1) @ v
            report("cfg3_global_attention_b63_h15_vs_fp64", o.view(64, 4096, H, 64)[63, :, H - 1], ref.float(), rtol=0, atol=2e-2 * float(ref.abs().max()))


# ======================================================================================================
# row kernels
# ======================================================================================================
@pytest.mark.parametrize("C", [4, 16, 64, 256, 768, 1024, 1152])
@pytest.mark.parametrize("TI,TO", [(F32, F32), (F32, BF16), (BF16, F32)])
def test_layernorm(C, TI, TO):
    ops, _ = _ops()
    rng = np.random.default_rng(C)
    x = torch.from_numpy(rng.standard_normal((37, C), dtype=np.float32) * 3 + 1).to(TI)
    w = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(C, dtype=np.float32))
    ref = torch.nn.functional.layer_norm(x.float(), (C,), w, b, 1e-6)
    out = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, out_dtype=TO)
    report(f"layernorm_C{C}_{TI}_{TO}", out, ref, **(dict(rtol=1e-4, atol=1e-4) if TO == F32 else dict(rtol=1e-2, atol=2e-2)))


def test_small_row_ops():
    ops, _ = _ops()
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((9, 768), dtype=np.float32))
    report("l2norm", ops.l2norm_rows(x.to(DEV)), torch.nn.functional.normalize(x, dim=-1), 1e-5, 1e-6)
    a = torch.from_numpy(rng.standard_normal((6, 64, 32), dtype=np.float32))
    b = torch.from_numpy(rng.standard_normal((64, 32), dtype=np.float32))
    report("add_periodic", ops.add(a.to(DEV), b.to(DEV)), a + b, 1e-6, 1e-6)
    report("add_periodic_bf16", ops.add(a.to(DEV), b.to(DEV), out_dtype=BF16), (a + b).to(BF16), 1e-2, 1e-2)
    report("cast_bf16", ops.cast(x.to(DEV), BF16), x.to(BF16), 0, 0)
    t = ops.tokens_to_nchw(a.to(DEV), 6, 64, 32)
    report("tokens_to_nchw", t, a.permute(0, 2, 1), 0, 0)
    report("nchw_to_tokens", ops.nchw_to_tokens(t, F32).view(6, 64, 32), a, 0, 0)
    ids = torch.from_numpy(rng.integers(0, 100, (3, 64)))
    table = torch.from_numpy(rng.standard_normal((100, 32), dtype=np.float32))
    pos = torch.from_numpy(rng.standard_normal((64, 32), dtype=np.float32))
    report("embed_tokens", ops.embed_tokens(ids.to(DEV), table.to(DEV), pos.to(DEV)).view(3, 64, 32), table[ids] + pos, 0, 0)


@pytest.mark.parametrize("p,size", [(16, 64), (14, 56)])
@pytest.mark.parametrize("T", [F32, BF16])
def test_patchify(p, size, T):
    ops, _ = _ops()
    rng = np.random.default_rng(p)
    img = torch.from_numpy(rng.standard_normal((2, 3, size, size), dtype=np.float32))
    K = 3 * p * p
    Kp = (K + 15) // 16 * 16
    g = size // p
    ref = img.reshape(2, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(2 * g * g, K)
    ref = torch.nn.functional.pad(ref, (0, Kp - K)).to(T)
    report(f"patchify_p{p}_{T}", ops.patchify(img.to(DEV), p, Kp, T), ref, 0, 0)


@pytest.mark.parametrize("C,T", [(64, BF16), (64, F32), (48, F32)])
def test_upscale_shuffle(C, T):
    """ConvTranspose2d(k=2,s=2) pixel shuffle of the GEMM output + bias + LayerNorm2d + GELU (mask_decoder.py:54-60): the C = 64
    four-pixels-per-wave kernel and the generic one against torch."""
    ops, _ = _ops()
    rng = np.random.default_rng(C)
    B, H, W = 2, 5, 7
    y = torch.from_numpy(rng.standard_normal((B * H * W, 4 * C), dtype=np.float32)).to(T)
    bias = torch.from_numpy(rng.standard_normal(C, dtype=np.float32))
    lw = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)); lb = torch.from_numpy(rng.standard_normal(C, dtype=np.float32))
    t = y.float().view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C) + bias     # [b, 2y+dy, 2x+dx, c]
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(t, (C,), lw, lb, 1e-6)).reshape(-1, C)
    out = ops.upscale_shuffle(y.to(DEV), B, H, W, C, bias=bias.to(DEV), ln_w=lw.to(DEV), ln_b=lb.to(DEV), eps=1e-6, act=1, out_dtype=F32)
    report(f"upscale_shuffle_C{C}_{T}", out, ref, 1e-4, 1e-4)


@pytest.mark.parametrize("C", [8, 6, 96])
def test_dwconv7x7(C):
    """Depthwise 7x7, pad 3, channels-last (mask_adapter.py:197-199 ConvNeXt block): the 4-channel vector path (C % 4 == 0) and
    the scalar path against torch's grouped conv."""
    ops, _ = _ops()
    rng = np.random.default_rng(C)
    B, H, W = 2, 9, 11
    x = torch.from_numpy(rng.standard_normal((B, H, W, C), dtype=np.float32))
    w = torch.from_numpy(rng.standard_normal((C, 1, 7, 7), dtype=np.float32) * 0.2)
    b = torch.from_numpy(rng.standard_normal(C, dtype=np.float32))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    w_t = w.reshape(C, 49).t().contiguous()                                       # [49, C]
    out = ops.dwconv7x7(x.to(DEV).reshape(-1, C), w_t.to(DEV), b.to(DEV), B, H, W)
    report(f"dwconv7x7_C{C}", out.view(B, H, W, C), ref, 1e-5, 1e-5)


def test_im2col3x3():
    ops, _ = _ops()
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((2, 8, 8, 16), dtype=np.float32))
    cols = torch.nn.functional.unfold(x.permute(0, 3, 1, 2), 3, padding=1)           # [B, C*9, HW], k = c*9 + tap
    ref = cols.reshape(2, 16, 9, 64).permute(0, 3, 2, 1).reshape(128, 144)           # -> [B*HW, tap*C + c]
    report("im2col3x3", ops.im2col3x3(x.to(DEV), 2, 8, 8), ref, 0, 0)
    # 16-byte raw-copy form (bf16, C = 256: the neck's 3x3 conv) and the element form (C = 4: row bytes not a multiple of 16)
    for (B, H, W, C, T) in ((3, 13, 9, 256, BF16), (2, 7, 5, 4, F32), (1, 64, 64, 256, BF16)):
        x = torch.from_numpy(rng.standard_normal((B, H, W, C), dtype=np.float32)).to(T)
        cols = torch.nn.functional.unfold(x.float().permute(0, 3, 1, 2), 3, padding=1)
        ref = cols.reshape(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C).to(T)
        report(f"im2col3x3_{B}x{H}x{W}x{C}_{T}", ops.im2col3x3(x.to(DEV), B, H, W), ref, 0, 0)


# ======================================================================================================
# attention
# ======================================================================================================
@pytest.mark.parametrize("hd,H,Tq,Tk", [(16, 8, 6, 4096), (16, 8, 4096, 6), (32, 8, 6, 6), (64, 12, 576, 576), (64, 12, 64, 64),
                                         (72, 16, 100, 729), (64, 2, 200, 333), (80, 2, 70, 130), (72, 16, 729, 729), (80, 16, 300, 257),
                                         (72, 3, 64, 64)])
@pytest.mark.parametrize("T", [F32, BF16])
def test_attention_plain(hd, H, Tq, Tk, T):
    """bf16 with head_dim 64 / 72 (SigLIP SO400M/14, the factory default) / 80 must run the MFMA flash kernel (asserted through
    cor_attention_kernel_id), fp32 the exact row-per-lane kernel."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    kid = nat.load().cor_attention_kernel_id(nat.BF16 if T == BF16 else nat.F32, hd, Tq, Tk, -1, 0)
    if T == BF16 and hd in (64, 72, 80) and Tq >= 64 and Tk >= 64:
        assert kid == nat.KERNEL_FLASH_MFMA, kid
    else:
        assert kid in (nat.KERNEL_ROWLANE, nat.KERNEL_FEWQ), kid
    rng = np.random.default_rng(hd + Tq + Tk)
    B, D = 2, H * hd
    q = torch.from_numpy(rng.standard_normal((B * Tq, D), dtype=np.float32)).to(T)
    kv = torch.from_numpy(rng.standard_normal((B * Tk, 2 * D), dtype=np.float32)).to(T)
    scale = hd ** -0.5
    sp = lambda t, Tn: t.float().reshape(B, Tn, H, hd).transpose(1, 2)
    a = torch.softmax(sp(q, Tq) @ sp(kv[:, :D], Tk).transpose(2, 3) * scale, -1) @ sp(kv[:, D:], Tk)
    ref = a.transpose(1, 2).reshape(B * Tq, D)
    kvd = kv.to(DEV)
    out = ops.attention(q.to(DEV), kvd[:, :D], kvd[:, D:], B, H, Tq, Tk, hd, scale, out_dtype=F32)
    report(f"attention_hd{hd}_{Tq}x{Tk}_{T}", out, ref, **(dict(rtol=1e-4, atol=1e-4) if T == F32 else dict(rtol=2e-2, atol=2e-2)))


@pytest.mark.parametrize("T", [F32, BF16])
def test_attention_online_softmax_rescale(T):
    """Forces the running-max rescale: a few keys late in the sequence dominate some queries (logit spikes of +30..60
    planted at chosen tiles), so the accumulator must be rescaled after many tiles were already summed."""
    ops, _ = _ops()
    rng = np.random.default_rng(77)
    B, H, hd, Tq, Tk = 1, 2, 64, 256, 640
    D = H * hd
    q = torch.from_numpy(rng.standard_normal((B * Tq, D), dtype=np.float32))
    k = torch.from_numpy(rng.standard_normal((B * Tk, D), dtype=np.float32))
    v = torch.from_numpy(rng.standard_normal((B * Tk, D), dtype=np.float32))
    for qi, ki, gain in ((3, 70, 6.0), (100, 333, 8.0), (255, 639, 10.0), (17, 5, 7.0), (64, 575, 9.0)):
        k[ki, :hd] = q[qi, :hd] * gain / 8.0          # q.k/sqrt(64) ~ gain * |q|^2 / 64 ~ gain * 8
        k[ki, hd:] = q[qi, hd:] * gain / 8.0
    q, k, v = q.to(T), k.to(T), v.to(T)
    sp = lambda t, Tn: t.float().reshape(B, Tn, H, hd).transpose(1, 2)
    ref = (torch.softmax(sp(q, Tq) @ sp(k, Tk).transpose(2, 3) * hd ** -0.5, -1) @ sp(v, Tk)).transpose(1, 2).reshape(B * Tq, D)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, H, Tq, Tk, hd, hd ** -0.5, out_dtype=F32)
    report(f"attention_rescale_{T}", out, ref, **(dict(rtol=1e-4, atol=1e-4) if T == F32 else dict(rtol=2e-2, atol=2e-2)))


@pytest.mark.parametrize("T", [BF16, F32])
def test_decoder_cross_attention_small_grid_kernel_is_bit_identical(T):
    """The decoder's token -> image attention (6 queries x 4096 keys, 8 heads x 16; transformer.py:163-166): below 128 (batch, head)
    blocks cor_attention runs attn_fewq_wide (one thread per (query, key slot): batch 1 is latency-bound), from 128 on attn_fewq (one thread
    per key slot, all queries): the same slots, the same merge order -> a sample's output is bit-identical alone and inside a large batch;
    both against torch fp32."""
    ops, _ = _ops()
    g = torch.Generator(device=DEV).manual_seed(11)
    B, H, hd, Tq, Tk = 20, 8, 16, 6, 4096
    q = torch.randn((B * Tq, H * hd), generator=g, device=DEV).to(T)
    k = torch.randn((B * Tk, H * hd), generator=g, device=DEV).to(T)
    v = torch.randn((B * Tk, H * hd), generator=g, device=DEV).to(T)
    big = ops.attention(q, k, v, B, H, Tq, Tk, hd, hd ** -0.5, out_dtype=F32)                      # 160 blocks: attn_fewq
    for b in (0, 7, 19):
        one = ops.attention(q[b * Tq:(b + 1) * Tq].contiguous(), k[b * Tk:(b + 1) * Tk].contiguous(), v[b * Tk:(b + 1) * Tk].contiguous(),
                            1, H, Tq, Tk, hd, hd ** -0.5, out_dtype=F32)                             # 8 blocks: attn_fewq_wide
        assert torch.equal(one, big[b * Tq:(b + 1) * Tq]), b
    sp = lambda t, Tn: t.float().reshape(B, Tn, H, hd).transpose(1, 2)
    ref = (torch.softmax(sp(q, Tq) @ sp(k, Tk).transpose(2, 3) * hd ** -0.5, -1) @ sp(v, Tk)).transpose(1, 2).reshape(B * Tq, H * hd)
    report(f"decoder_cross_attention_{T}", big, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["win14", "glob16"])
def test_sam_attention_vs_reference_golden(tag):
    """qkv GEMM -> cor_sam_attention -> proj GEMM against lib/sam_model/image_encoder.py Attention (golden)."""
    ops, engine = _ops()
    g = load(f"sam_attention_{tag}")
    S, dim, heads = int(g["S"]), int(g["dim"]), int(g["heads"])
    cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,), window=14, img=S * 16, patch=16, out=16)
    spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.attn.")}
    sd = dev(ocfg.random_state(spec, int(g["seed_params"])))
    x = make_inputs(int(g["seed_inputs"]), x=(3, S, S, dim))["x"].to(DEV).reshape(-1, dim)
    p = "e.blocks.0.attn."
    qkv = ops.gemm(x, sd[p + "qkv.weight"], bias=sd[p + "qkv.bias"])
    a = ops.sam_attention(qkv, sd[p + "qkv.bias"], sd[p + "rel_pos_h"], sd[p + "rel_pos_w"], 3, heads, S, 0)
    y = ops.gemm(a, sd[p + "proj.weight"], bias=sd[p + "proj.bias"])
    report(f"sam_attention_golden_{tag}", y.view(3, S, S, dim), g["y"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("B,H,amp", [(1, 12, 1.0), (3, 16, 1.0), (2, 12, 6.0), (2, 5, 0.05), (32, 12, 1.0)])
def test_global_attention_pipelined_vs_fp64_reference_and_chain_kernel(B, H, amp):
    """flash_global_pipe (software-pipelined, LDS-DMA rings, lazy rescale; default) at T = 4096 keys against
    (a) an fp64 CPU evaluation of the reference formula (image_encoder.py:225-241,326-362, oracle.sam.rel_pos_bias, pinned by the
        goldens) on the SAME bf16 operands, for three (image, head) pairs of the launch incl. the last image of the batch-32 case
        (the benchmarked batch) - not a self-comparison;
    (b) flash_fwd<1> (one tile at a time, exact running max) on the whole tensor: same math, different rounding points.
    amp = 6 drives score ranges of +-100 log2 units (lazy rescale fires often, strong peaks), amp = 0.05 nearly uniform attention.
    Also the q_prescale path the engine runs (scale * log2 e folded into the q third by the caller), both kernel forms."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    g = torch.Generator(device=DEV).manual_seed(B * 100 + H)
    d = H * 64
    qkv = (torch.randn((B * 4096, 3 * d), generator=g, device=DEV) * amp).to(BF16)
    pad = torch.randn((3 * d,), generator=g, device=DEV).to(BF16)
    rh = torch.randn((127, 64), generator=g, device=DEV) * 0.3
    rw = torch.randn((127, 64), generator=g, device=DEV) * 0.3
    out = ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, out_dtype=F32)
    assert torch.equal(out, ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, out_dtype=F32))     # run-to-run reproducible
    chain = ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, out_dtype=F32, variant=1)            # chain form (per-call choice)
    c = nat.Q_PRESCALE_HD64
    qs, ps = qkv.float(), pad.float()
    qs[:, :d] *= c; ps[:d] *= c
    qs, ps = qs.to(BF16), ps.to(BF16)
    o_p = ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c)             # what the engine runs
    assert torch.equal(o_p, ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c))
    o_c = ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c, variant=1)
    # round 5: with a pre-scaled q the default folds the column bias into the score MFMA's accumulator start value; variant 2 = the
    # fma form (round 4's default), variant 4 = flash_global_w64 (64 queries per wave, one wave per SIMD, asm MFMAs, deferred rescale)
    o_f = ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c, variant=2)
    o_w = ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c, variant=4)
    assert torch.equal(o_w, ops.sam_attention(qs, ps, rh, rw, B, H, 64, 0, out_dtype=F32, q_prescale=c, variant=4))
    hard = amp > 2
    # (a) fp64 reference on sampled (image, head) pairs
    x = qkv.view(B, 4096, 3, H, 64)
    rhd, rwd = rh.to(BF16).double().cpu(), rw.to(BF16).double().cpu()                            # the kernels feed the tables to the MFMA in bf16
    pairs = sorted({(0, 0), (B - 1, H - 1), (B // 2, H // 2)})
    worst = {}
    for (b, hh) in pairs:
        q, k, v = (x[b, :, j, hh].double().cpu() for j in range(3))
        bias = osam.rel_pos_bias(q[None], rhd, rwd, 64)[0]
        ref = torch.softmax((q * 0.125) @ k.T + bias, dim=-1) @ v
        scale = float(ref.abs().max())
        for name, got, tol, rl in (("pipe", out, 2e-2, 4e-3), ("chain", chain, 2e-2, 4e-3), ("pipe_prescaled", o_p, 8e-2 if hard else 2e-2, 1.5e-2 if hard else 5e-3),
                                   ("chain_prescaled", o_c, 8e-2 if hard else 2e-2, 1.5e-2 if hard else 5e-3),
                                   ("pipe_fma_prescaled", o_f, 8e-2 if hard else 2e-2, 1.5e-2 if hard else 5e-3),
                                   ("w64_prescaled", o_w, 8e-2 if hard else 2e-2, 1.5e-2 if hard else 5e-3)):
            r = report(f"global_attn_{name}_vs_fp64_B{B}_H{H}_amp{amp}_b{b}_h{hh}", got.view(B, 4096, H, 64)[b, :, hh], ref.float(), rtol=0, atol=tol * scale)
            assert r["rel_l2"] <= rl * max(1.0, amp if not hard else 1.0), (name, b, hh, r)
            worst[name] = max(worst.get(name, 0.0), r["rel_l2"])
    _note(name=f"global_attn_vs_fp64_relL2_B{B}_H{H}_amp{amp}", pairs=len(pairs), **worst)
    # (b) the two kernel forms against each other on the whole tensor
    report(f"global_attn_pipe_vs_chain_B{B}_H{H}_amp{amp}", out, chain, rtol=2e-2, atol=2e-2 * float(chain.abs().max()))
    tol = 8e-2 if hard else 2e-2
    r_p = report(f"global_attn_prescaled_vs_chain_B{B}_H{H}_amp{amp}", o_p, chain, rtol=tol, atol=tol * float(chain.abs().max()))
    assert r_p["rel_l2"] <= (2e-2 if hard else 5e-3), r_p
    for name, got in (("fma", o_f), ("w64", o_w)):
        r_v = report(f"global_attn_prescaled_{name}_vs_chain_B{B}_H{H}_amp{amp}", got, chain, rtol=tol, atol=tol * float(chain.abs().max()))
        assert r_v["rel_l2"] <= (2e-2 if hard else 5e-3), r_v


def test_production_library_rejects_probe_and_experimental_selectors():
    """The timing probes (cycle counters INSTEAD of outputs) and the result-destroying GEMM ablation bits of round 2 live in the
    COR_PROBES build only (make -C cor_amd/csrc probes): the production ABI answers COR_EINVAL instead of silent garbage."""
    from cor_amd import _native as nat
    ops, _ = _ops()
    qkv = torch.zeros((4096, 3 * 64), device=DEV, dtype=BF16)
    rel = torch.zeros((127, 64), device=DEV)
    for variant in (3, 5, 6, 9, 10, 14, 16, 17, 77):
        with pytest.raises(nat.NativeError):
            ops.sam_attention(qkv, None, rel, rel, 1, 1, 64, 0, variant=variant)
    a = torch.zeros((256, 128), device=DEV, dtype=BF16)
    for cfg in (1 << 8, 2 << 8, 4 << 8, 8 << 8, 13 | (1 << 8), 14, 5, 200):
        with pytest.raises(nat.NativeError):
            ops.gemm(a, a, cfg=cfg)
    assert ops.gemm(a, a, cfg=2).shape == (256, 256) and ops.gemm(a, a, cfg=13).shape == (256, 256)


@pytest.mark.parametrize("B,H,grid,amp,TO", [(1, 12, 64, 1.0, BF16), (3, 16, 64, 1.0, F32), (2, 5, 64, 6.0, BF16), (2, 12, 64, 0.05, BF16),
                                               (2, 3, 28, 1.0, BF16), (1, 2, 30, 2.0, F32)])
def test_windowed_attention_block_per_window_vs_chain_kernel(B, H, grid, amp, TO):
    """win_attn (one 7-wave block per (window, head): bias and running reference folded into the score accumulator, Q
    pre-scaled in bf16; default) against flash_fwd<2> (round-1 chain form, scale applied in fp32) on the same inputs, and
    both against an fp64 CPU evaluation of image_encoder.py:225-241,244-290,326-362 on the SAME bf16 operands. Grids 64 (5x5
    windows, bottom/right padded), 28 (exact 2x2), 30 (3x3 with 12-token padding); amp = 6 drives log2-scores to +-100 (the
    lazy rescale fires), amp = 0.05 nearly uniform attention. Padded tokens use pad_row (the qkv bias), as the reference's
    zero-padding after norm1 implies."""
    ops, _ = _ops()
    g = torch.Generator(device=DEV).manual_seed(B * 100 + H + grid)
    d = H * 64
    qkv = (torch.randn((B * grid * grid, 3 * d), generator=g, device=DEV) * amp).to(BF16)
    pad = (torch.randn((3 * d,), generator=g, device=DEV) * amp).to(BF16)
    rh = torch.randn((27, 64), generator=g, device=DEV) * 0.3
    rw = torch.randn((27, 64), generator=g, device=DEV) * 0.3
    out = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=TO)
    out2 = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=TO)
    old = ops.sam_attention(qkv, pad, rh, rw, B, H, grid, 14, out_dtype=TO, variant=1)
    assert torch.equal(out, out2)                                   # run-to-run reproducible
    # fp64 reference on the same bf16 operands (tables rounded to bf16 as both kernels feed them to the MFMA)
    nW = (grid + 13) // 14
    P = nW * 14
    x = pad.double().cpu().repeat(B, P, P, 1)
    x[:, :grid, :grid] = qkv.double().cpu().view(B, grid, grid, 3 * d)
    xw = x.view(B, nW, 14, nW, 14, 3, H, 64).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nW * nW, 3, H, 196, 64)
    q, k, v = xw[:, 0], xw[:, 1], xw[:, 2]
    qf_, kf_, vf_ = (t.reshape(-1, 196, 64) for t in (q, k, v))
    bias = osam.rel_pos_bias(qf_, rh.to(BF16).double().cpu(), rw.to(BF16).double().cpu(), 14)      # oracle (pinned by the goldens)
    att = (torch.softmax((qf_ * 0.125) @ kf_.transpose(1, 2) + bias, dim=-1) @ vf_).reshape(-1, H, 196, 64)
    ref = att.reshape(B, nW, nW, H, 14, 14, 64).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, P, P, d)[:, :grid, :grid].reshape(-1, d).float()
    scale = float(ref.abs().max())
    # Budgets. The chain form multiplies exact bf16 products by scale*log2e in fp32; win_attn feeds the MFMA q * scale*log2e
    # ROUNDED to bf16 (one more 2^-9 rounding on the q side). At amp <= 2 (log2-scores within +-25) both sit at ~1e-3 rel-L2;
    # at amp = 6 the softmax is nearly one-hot, |score| ~ 100 log2 units, the extra rounding moves a score by ~0.1 and a
    # near-tie between the two best keys by up to ~8 % of the value range: max-abs budget 8e-2 of the scale there (chain form
    # 2e-2), rel-L2 <= 1.5e-2.
    hard = amp > 2
    tol_new = (8e-2 if hard else 2e-2) if TO == BF16 else (8e-2 if hard else 1.2e-2)
    tol_old = 2e-2 if TO == BF16 else 1.2e-2
    r_old = report(f"win_attn_old_B{B}_H{H}_g{grid}_amp{amp}_{TO}", old, ref, rtol=0, atol=tol_old * scale)
    r_new = report(f"win_attn_new_B{B}_H{H}_g{grid}_amp{amp}_{TO}", out, ref, rtol=0, atol=tol_new * scale)
    _note(name=f"win_attn_relL2_B{B}_H{H}_g{grid}_amp{amp}_{TO}", new=r_new["rel_l2"], old=r_old["rel_l2"])
    assert r_new["rel_l2"] <= (1.5e-2 if hard else 4e-3 * max(1.0, amp)), (r_new, r_old)
    # q_prescale path (what the engine runs: scale * log2 e folded into the q third by the caller), both kernel forms
    from cor_amd import _native as nat
    c = nat.Q_PRESCALE_HD64
    qs, ps = qkv.float(), pad.float()
    qs[:, :d] *= c; ps[:d] *= c
    qs, ps = qs.to(BF16), ps.to(BF16)
    for variant in (0, 1):
        o_p = ops.sam_attention(qs, ps, rh, rw, B, H, grid, 14, out_dtype=TO, variant=variant, q_prescale=c)
        r_p = report(f"win_attn_prescaled_v{variant}_B{B}_H{H}_g{grid}_amp{amp}_{TO}", o_p, ref, rtol=0, atol=tol_new * scale)
        assert r_p["rel_l2"] <= (1.5e-2 if hard else 4e-3 * max(1.0, amp)), r_p


@pytest.mark.parametrize("window", [14, 0])
@pytest.mark.parametrize("T,hd", [(F32, 64), (BF16, 64), (BF16, 80)])
def test_sam_attention_real_dims(window, T, hd):
    """hd 64 (SAM-B/L) and 80 (SAM-H), grid 64 (window 14 -> 5x5 padded windows; 0 -> global 4096 keys) against the oracle.
    bf16 must run on the matrix cores for both head dims (cor_attention_kernel_id)."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    heads, dim, grid, B = 2, 2 * hd, 64, 1
    kid = nat.load().cor_attention_kernel_id(nat.BF16 if T == BF16 else nat.F32, hd, 0, 0, window, grid)
    if T == BF16:
        assert kid in (nat.KERNEL_FLASH_MFMA, nat.KERNEL_FLASH_PIPELINED, nat.KERNEL_WINDOW_BLOCK), kid
    else:
        assert kid == nat.KERNEL_ROWLANE
    cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,) if window == 0 else (), window=14, img=1024, patch=16, out=16)
    spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.attn.")}
    sd = ocfg.random_state(spec, 31)
    p = "e.blocks.0.attn."
    x = make_inputs(32, x=(B, grid, grid, dim))["x"]
    if T == BF16:   # same rounded operands on both sides
        x = x.to(BF16).float()
        for k in (p + "qkv.weight", p + "proj.weight"):
            sd[k] = sd[k].to(BF16).float()
    if window:
        xw, pad_hw = osam.to_windows(x, window)
        ref = osam.from_windows(osam.vit_attention(sd, p, xw, heads), window, pad_hw, (grid, grid))
    else:
        ref = osam.vit_attention(sd, p, x, heads)
    d = dev(sd)
    xd = x.to(DEV).reshape(-1, dim).to(T)
    qkv = ops.gemm(xd, d[p + "qkv.weight"].to(T), bias=d[p + "qkv.bias"])
    a = ops.sam_attention(qkv, d[p + "qkv.bias"].to(T), d[p + "rel_pos_h"], d[p + "rel_pos_w"], B, heads, grid, window)
    y = ops.gemm(a, d[p + "proj.weight"].to(T), out_dtype=F32, bias=d[p + "proj.bias"])
    report(f"sam_attention_hd{hd}_w{window}_{T}", y.view(B, grid, grid, dim), ref,
           **(dict(rtol=1e-3, atol=2e-4) if T == F32 else dict(rtol=5e-2, atol=5e-2)))


# ======================================================================================================
# stages against REFERENCE golden vectors (fp32 exact mode)
# ======================================================================================================
@pytest.mark.parametrize("tag", ["img256", "img1024"])
def test_sam_encoder_vs_reference_golden(tag):
    _, engine = _ops()
    g = load(f"sam_encoder_{tag}")
    cfg = dict(dim=int(g["dim"]), heads=int(g["heads"]), depth=int(g["depth"]), global_idx=tuple(int(i) for i in g["global_idx"]),
               window=14, img=int(g["img"]), patch=16, out=int(g["out"]))
    pk = packer(ocfg.random_state(ocfg.sam_encoder_spec(cfg), int(g["seed_params"])), F32)
    engine.pack_sam_encoder(pk, cfg)
    B = int(g["B"])
    x = make_inputs(int(g["seed_inputs"]), x=(B, 3, cfg["img"], cfg["img"]))["x"].to(DEV)
    tok = engine.sam_encoder(pk.W, x, cfg, F32)
    gr = cfg["img"] // 16
    from cor_amd import ops
    y = ops.tokens_to_nchw(tok, B, gr * gr, cfg["out"]).view(B, cfg["out"], gr, gr)
    report(f"sam_encoder_golden_{tag}", y, g["y"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("mode", [F32, BF16])
@pytest.mark.parametrize("B", [1, 5, 32])
def test_decoder_heads_one_launch_equals_the_gemm_path(mode, B):
    """cor_decoder_heads (the decoder's five output MLPs in one launch) against the 15-GEMM path it replaces, same packed weights and
    tokens: fp32 within 2e-5 of the scale (fp32 FMA chains vs the exact-fp32 MFMA), bf16 within the rounding of the hidden activations;
    a sample's result does not depend on the batch it is in (bitwise)."""
    ops, engine = _ops()
    pk = packer(ocfg.random_state(dict(ocfg.mask_decoder_spec(), **ocfg.prompt_encoder_spec()), 3), mode)
    engine.pack_mask_decoder(pk)
    W = pk.W
    gen = torch.Generator().manual_seed(B)
    hs = (torch.randn((B * 6, 256), generator=gen) * 2).to(mode).to(DEV)
    hyper, iou = ops.decoder_heads(hs, W["mask_decoder.heads.w01"], W["mask_decoder.heads.b01"], W["mask_decoder.heads.w2"], W["mask_decoder.heads.b2"])
    hs3 = hs.view(B, 6 * 256)
    p = "mask_decoder."
    ref_h = torch.empty((B, 4, 32), dtype=F32, device=DEV)
    for i in range(4):
        m = f"{p}output_hypernetworks_mlps.{i}.layers."
        a = engine._lin(W, m + "0.", hs3[:, (1 + i) * 256:(2 + i) * 256], mode, act=engine.ACT_RELU)
        a = engine._lin(W, m + "1.", a, mode, act=engine.ACT_RELU)
        engine._lin(W, m + "2.", a, F32, out=ref_h.view(B, 128)[:, 32 * i:32 * (i + 1)])
    m = p + "iou_prediction_head.layers."
    a = engine._lin(W, m + "0.", hs3[:, 0:256], mode, act=engine.ACT_RELU)
    a = engine._lin(W, m + "1.", a, mode, act=engine.ACT_RELU)
    ref_i = engine._lin(W, m + "2.", a, F32)
    tol = 2e-5 if mode == F32 else 2e-2
    for got, ref, name in ((hyper, ref_h, "hyper"), (iou, ref_i, "iou")):
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        _note(name=f"decoder_heads_{name}", mode=str(mode), B=B, max_abs_err=err, scale=scale)
        assert err <= tol * max(scale, 1.0), (name, err, scale)
    one_h, one_i = ops.decoder_heads(hs[6 * (B - 1):].contiguous(), W["mask_decoder.heads.w01"], W["mask_decoder.heads.b01"], W["mask_decoder.heads.w2"], W["mask_decoder.heads.b2"])
    assert torch.equal(one_h[0], hyper[B - 1]) and torch.equal(one_i[0], iou[B - 1])


def test_mask_decoder_vs_reference_golden():
    ops, engine = _ops()
    g = load("mask_decoder")
    pk = packer(ocfg.random_state(dict(ocfg.mask_decoder_spec(), **ocfg.prompt_encoder_spec()), int(g["seed_params"])), F32)
    engine.pack_mask_decoder(pk)
    engine.pack_prompt_encoder(pk)
    inp = make_inputs(int(g["seed_inputs"]), emb=(2, 256, 64, 64), sparse=(2, 1, 256))
    pe = ops.tokens_to_nchw(pk.W["prompt.dense_pe"], 1, 4096, 256).view(1, 256, 64, 64)
    report("dense_pe_golden", pe[..., ::4, ::4], g["dense_pe"], rtol=1e-4, atol=1e-4)
    emb_tok = ops.nchw_to_tokens(inp["emb"].to(DEV), F32)
    feat = inp["sparse"].to(DEV).reshape(2, 256).contiguous()
    for mm in (0, 1):
        final, iou, best, masks_all, keys = engine.mask_decoder(pk.W, emb_tok, feat, F32, bool(mm), all_masks=True)
        sl = slice(1, None) if mm else slice(0, 1)
        report(f"decoder_masks_golden_mm{mm}", masks_all[:, sl][..., ::4, ::4], g[f"masks_{mm}"], rtol=1e-3, atol=1e-3)
        report(f"decoder_iou_golden_mm{mm}", iou[:, sl], g[f"iou_{mm}"], rtol=1e-3, atol=1e-3)
        # the selected mask equals the argmax-IoU slice of the full set (sam_with_sup_branch.py:96-100)
        ref_best = torch.from_numpy(g[f"iou_{mm}"]).argmax(1)
        assert torch.equal(best.cpu(), ref_best), (best, ref_best)
        k_off = 1 if mm else 0
        pick = masks_all[torch.arange(2), k_off + best].unsqueeze(1)
        report(f"decoder_final_is_selected_mm{mm}", final, pick, rtol=0, atol=0)
    src = ops.tokens_to_nchw(keys, 2, 4096, 256).view(2, 256, 64, 64)
    report("decoder_src_golden", src[..., ::4, ::4], g["src"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("D,grid", [(768, 24), (1024, 24), (1152, 27)])
def test_mask_adapter_vs_reference_golden(D, grid):
    ops, engine = _ops()
    g = load(f"mask_adapter_D{D}")
    pk = packer(ocfg.random_state(ocfg.mask_adapter_spec(D, "mp."), int(g["seed_params"])), F32)
    engine.pack_mask_adapter(pk, "mp.")
    inp = make_inputs(int(g["seed_inputs"]), feat=(2, D, grid, grid), mask=("mask", 2, 384))
    report(f"bilinear_golden_{grid}", ops.bilinear(inp["mask"].to(DEV), grid, grid), g["mask_small"], 1e-5, 1e-6)
    feat_tok = ops.nchw_to_tokens(inp["feat"].to(DEV), F32)
    pooled, maps = engine.mask_adapter_pooling(pk.W, feat_tok, inp["mask"].to(DEV), 2, grid, D, F32, "mp.")
    report(f"adapter_maps_golden_D{D}", ops.tokens_to_nchw(maps, 2, grid * grid, 8).view(2, 8, grid, grid), g["maps"], 1e-3, 1e-3)
    report(f"adapter_pooled_golden_D{D}", pooled.view(2, 1, D), g["y"], 1e-3, 1e-4)


def test_masked_pooling_and_region_embedding_vs_reference_golden():
    ops, _ = _ops()
    g = load("masked_pooling")
    inp = make_inputs(int(g["seed_inputs"]), feat=(2, 768, 24, 24), mask=("mask", 2, 384))
    m = ops.bilinear(inp["mask"].to(DEV), 24, 24)
    report("masked_pooling_golden", ops.masked_pool(ops.nchw_to_tokens(inp["feat"].to(DEV), F32), m, 2, 576, 768), g["y"], 1e-4, 1e-5)
    g = load("region_embedding")
    inp = make_inputs(int(g["seed_inputs"]), emb=(3, 256, 64, 64), mask=("mask", 3, 256), feat=(3, 1, 256))
    from cor_amd import retrieval
    r = retrieval.region_embedding(inp["emb"].to(DEV), inp["mask"].to(DEV))
    report("region_embedding_golden", r, g["y"], 1e-4, 1e-5)


@pytest.mark.parametrize("D", [768, 1024])
def test_cir_fuse_vs_reference_golden(D):
    ops, engine = _ops()
    g = load(f"cir_fuse_D{D}")
    pk = packer(ocfg.random_state(ocfg.fuse_spec(D, "f."), int(g["seed_params"])), F32)
    engine.pack_fuse(pk, "f.")
    inp = make_inputs(int(g["seed_inputs"]), img=(4, D), txt=(4, D))
    img, txt = inp["img"].to(DEV), inp["txt"].to(DEV)
    raw = torch.cat([img, txt], 1).contiguous()   # test-side plumbing only
    gate = lambda n, x: ops.gemm(ops.gemm(x, pk.W[f"f.{n}.0.weight"], bias=pk.W[f"f.{n}.0.bias"], act=ops.ACT_RELU),
                                 pk.W[f"f.{n}.3.weight"], bias=pk.W[f"f.{n}.3.bias"], act=ops.ACT_SIGMOID)
    cat = ops.fuse_gate(img, txt, gate("atten_Image", raw), gate("atten_Text", raw))
    dyn = gate("dynamic_scalar", cat)
    report(f"cir_fuse_dyn_golden_D{D}", dyn, g["dyn"], 1e-3, 1e-4)
    report(f"cir_fuse_golden_D{D}", ops.fuse_mix(cat, dyn), g["y"], 1e-3, 1e-5)


# ======================================================================================================
# whole model
# ======================================================================================================
def _build(sam_depth, gidx, gcfg, pooling):
    from cor_amd.lib.sam_model.image_encoder import ImageEncoderViT
    from cor_amd.lib.sam_model.mask_decoder import MaskDecoder
    from cor_amd.lib.sam_model.my_prompt_encoder import PromptEncoder
    from cor_amd.lib.sam_model.transformer import TwoWayTransformer
    from cor_amd.lib.sam_with_sup_branch import CirSegModelWithQuerySupportFeat
    from cor_amd.lib.support_branch import SupportBranch
    return CirSegModelWithQuerySupportFeat(
        image_encoder=ImageEncoderViT(embed_dim=768, depth=sam_depth, num_heads=12, global_attn_indexes=gidx),
        support_branch=SupportBranch("ViT-B-16-SigLIP-384", None, pooling, siglip_cfg=gcfg),
        prompt_encoder=PromptEncoder(256, (64, 64)),
        mask_decoder=MaskDecoder(transformer_dim=256, transformer=TwoWayTransformer(2, 256, 8, 2048)))


@pytest.mark.parametrize("pooling", ["MaskAdapterPooling", "MaskedPooling"])
def test_full_forward_vs_reference_golden(pooling):
    """SAM-B (all 12 blocks) + 2-block SigLIP stand-in: the golden outputs come from the reference's own
    build_model_with_query_support_feat(...).forward (tools/make_golden.py gen_toplevel). fp32 exact mode."""
    from cor_amd import config
    g = load(f"toplevel_{pooling}")
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    model = _build(12, (2, 5, 8, 11), gcfg, pooling)
    # the stand-in used for the golden run has no MAP head
    spec = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = ocfg.random_state({k: v for k, v in spec.items() if "attn_pool" not in k}, int(g["seed_params"]))
    missing = model.load_state_dict(sd, strict=False)
    assert all("attn_pool" in k for k in missing.missing_keys) and not missing.unexpected_keys
    assert sorted(sd) == [str(k) for k in g["keys"]]
    model = model.to(DEV).eval()
    inp = make_inputs(int(g["seed_inputs"]), q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512), mask=("mask", 1, 384))
    kw = dict(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV), change_text_inputs=inp["text"].to(DEV),
              support_mask_inputs=inp["mask"].to(DEV))
    for mm in (1, 0):
        masks, emb, feat = model(**kw, multimask_output=bool(mm))
        assert masks.shape == (1, 1, 256, 256) and emb.shape == (1, 256, 64, 64) and feat.shape == (1, 1, 256)
        report(f"full_masks_golden_{pooling}_mm{mm}", masks[..., ::4, ::4], g[f"masks_{mm}"], rtol=1e-3, atol=2e-3)
    report(f"full_emb_golden_{pooling}", emb[..., ::4, ::4], g["emb"], rtol=1e-3, atol=1e-3)
    report(f"full_feat_golden_{pooling}", feat, g["feat"], rtol=1e-3, atol=1e-4)
    # mask "argmax" parity = the >0 threshold of the logits (mask_threshold 0.0) on the sub-sampled golden
    agree = ((masks[..., ::4, ::4].cpu() > 0) == (torch.from_numpy(g["masks_0"]) > 0)).float().mean().item()
    assert agree == 1.0, f"mask threshold disagreement: {1 - agree:.2e}"


def test_full_forward_bf16_mode_vs_oracle():
    """bf16 fast mode (what the reference runs under accelerator.autocast): budget = 3e-2 of the output scale on the
    embeddings / support feature, mask sign agreement >= 99 %. B=2 exercises batching."""
    from cor_amd import config
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    model = _build(2, (1,), gcfg, "MaskAdapterPooling")
    sd = ocfg.random_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, 41)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    inp = make_inputs(42, q=(2, 3, 1024, 1024), s=(2, 3, 384, 384), text=("tokens", 2, 64, 512), mask=("mask", 2, 384))
    scfg = dict(model.image_encoder.cfg)
    ref_emb = osam.image_encoder(sd, inp["q"], scfg)
    ref_feat = osup.support_branch(sd, inp["s"], inp["text"], inp["mask"], gcfg, "MaskAdapterPooling")
    ref_masks, ref_iou, _ = osam.mask_decoder(sd, ref_emb, osam.dense_pe(sd), ref_feat, osam.dense_no_mask(sd, 2), True)
    kw = dict(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV), change_text_inputs=inp["text"].to(DEV),
              support_mask_inputs=inp["mask"].to(DEV))
    # fp32 exact mode, B=2
    masks, emb, feat, aux = model.forward_with_aux(**kw, multimask_output=True)
    report("full_fp32_emb_vs_oracle", emb, ref_emb, 1e-3, 2e-3)
    report("full_fp32_feat_vs_oracle", feat, ref_feat, 1e-3, 1e-4)
    report("full_fp32_masks_vs_oracle", aux["masks"][:, 1:], ref_masks, 2e-3, 5e-3)
    assert torch.equal(aux["best"].cpu(), ref_iou.argmax(1))
    # bf16 mode through autocast, like the reference's inference harness (vailder.py:416)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        masks_b, emb_b, feat_b = model(**kw, multimask_output=True)
    es = ref_emb.abs().max().item()
    report("full_bf16_emb_vs_oracle", emb_b, ref_emb, 0, 3e-2 * es)
    report("full_bf16_feat_vs_oracle", feat_b, ref_feat, 0, 3e-2)
    ref_final = ref_masks[torch.arange(2), ref_iou.argmax(1)].unsqueeze(1)
    agree = ((masks_b.cpu() > 0) == (ref_final > 0)).float().mean().item()
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(name="full_bf16_mask_sign_agreement", value=agree)) + "\n")
    assert agree >= 0.99, agree


def _note(**rec):
    """One JSON line into the parity report (counts quoted in DESIGN.md section 4) and onto stdout (pytest -s / -rA)."""
    print("PARITY-NOTE", json.dumps(rec), flush=True)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("pooling", ["MaskAdapterPooling", "MaskedPooling"])
def test_full_depth_bf16_vs_reference_golden_with_counts(pooling):
    """The BENCHMARKED mode (bf16) at FULL depth (all 12 SAM-B blocks) against the reference's own outputs
    (tests/golden/toplevel_*.npz = lib/sam_with_sup_branch.py:57-104 run by tools/make_golden.py), not a self-comparison.
    Stated budgets (bf16 operands, fp32 accumulation / residual stream / LayerNorm statistics / softmax):
      embeddings: rel-L2 <= 3e-2, max|err| <= 6e-2 of the output scale;   support feature: max|err| <= 3e-2, rel-L2 <= 3e-2;
      masks (raw logits): rel-L2 <= 6e-2; mask-sign flips (threshold 0.0 = the reference's mask_threshold) <= 1 % of pixels;
      IoU-argmax flips vs the fp32 exact mode (itself pinned to the golden): 0 of 1.
    Reference-anchored budget on top: every error (emb / feat / masks rel-L2, sign flips) is <= 1.5x the error of the REFERENCE run
    under bf16 autocast on the same parameters and inputs (tests/golden/toplevel_autocast_bf16_*.npz, tools/make_golden.py
    gen_toplevel_autocast: utils/vailder.py:416 runs inference under accelerator.autocast() bf16).
    Then retrieval with the bf16 feature vs the reference's fp32 feature on a planted 100k-row bf16 gallery:
    Recall@1 = 1.0 and the count of top-10 index mismatches is recorded."""
    from cor_amd import config, retrieval
    g = load(f"toplevel_{pooling}")
    ga = load(f"toplevel_autocast_bf16_{pooling}")       # the REFERENCE run under torch.autocast("cpu", bf16) on the same parameters / inputs
    assert int(ga["seed_params"]) == int(g["seed_params"]) and int(ga["seed_inputs"]) == int(g["seed_inputs"])
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    model = _build(12, (2, 5, 8, 11), gcfg, pooling)
    spec = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = ocfg.random_state({k: v for k, v in spec.items() if "attn_pool" not in k}, int(g["seed_params"]))
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).eval()
    inp = make_inputs(int(g["seed_inputs"]), q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512), mask=("mask", 1, 384))
    kw = dict(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV), change_text_inputs=inp["text"].to(DEV),
              support_mask_inputs=inp["mask"].to(DEV))
    _, _, feat32, aux32 = model.forward_with_aux(**kw, multimask_output=True)                 # fp32 exact mode: pinned to the golden
    flips_total, px_total = 0, 0
    for mm in (1, 0):
        with torch.autocast("cuda", dtype=torch.bfloat16):                                  # like vailder.py:416
            masks, emb, feat, aux = model.forward_with_aux(**kw, multimask_output=bool(mm))
        ref_m = torch.from_numpy(g[f"masks_{mm}"])
        got_m = masks[..., ::4, ::4].float().cpu()
        rel = float((got_m - ref_m).norm() / ref_m.norm())
        flips = int(((got_m > 0) != (ref_m > 0)).sum())
        flips_total += flips; px_total += ref_m.numel()
        # reference-anchored budget: the reference's OWN bf16-autocast outputs against its fp32 outputs, same sub-sampled pixels
        ac_m = torch.from_numpy(ga[f"masks_{mm}"])
        ac_rel = float((ac_m - ref_m).norm() / ref_m.norm())
        ac_flips = int(((ac_m > 0) != (ref_m > 0)).sum())
        _note(name=f"full_depth_bf16_masks_{pooling}_mm{mm}", rel_l2=rel, max_abs=float((got_m - ref_m).abs().max()),
              max_ref=float(ref_m.abs().max()), mask_sign_flips=flips, pixels=ref_m.numel(),
              reference_autocast_rel_l2=ac_rel, reference_autocast_sign_flips=ac_flips)
        assert rel <= 6e-2, rel
        assert rel <= 1.5 * ac_rel and flips <= 1.5 * ac_flips + 2, (rel, ac_rel, flips, ac_flips)
        if mm:
            argmax_flips = int((aux["best"].cpu() != aux32["best"].cpu()).sum())
            _note(name=f"full_depth_bf16_iou_argmax_{pooling}", argmax_flips=argmax_flips, of=int(aux["best"].numel()),
                  iou_bf16=aux["iou"].float().cpu().tolist(), iou_fp32=aux32["iou"].float().cpu().tolist())
            assert argmax_flips == 0
    assert flips_total <= 0.01 * px_total, (flips_total, px_total)
    es = float(np.abs(g["emb"]).max())
    r_e = report(f"full_depth_bf16_emb_{pooling}", emb[..., ::4, ::4], g["emb"], rtol=0, atol=6e-2 * es)
    r_f = report(f"full_depth_bf16_feat_{pooling}", feat, g["feat"], rtol=0, atol=3e-2)
    assert r_e["rel_l2"] <= 3e-2 and r_f["rel_l2"] <= 3e-2, (r_e, r_f)
    # ... and against the reference's own bf16 error on the same tensors (emb sub-sampled like the fixture, feat whole)
    ge, gf = torch.from_numpy(g["emb"]), torch.from_numpy(g["feat"])
    ac_e = float((torch.from_numpy(ga["emb"]) - ge).norm() / ge.norm())
    ac_f = float((torch.from_numpy(ga["feat"]) - gf).norm() / gf.norm())
    _note(name=f"full_depth_bf16_vs_reference_autocast_{pooling}", hip_emb_rel_l2=r_e["rel_l2"], reference_autocast_emb_rel_l2=ac_e,
          hip_feat_rel_l2=r_f["rel_l2"], reference_autocast_feat_rel_l2=ac_f)
    assert r_e["rel_l2"] <= 1.5 * ac_e and r_f["rel_l2"] <= 1.5 * ac_f, (r_e["rel_l2"], ac_e, r_f["rel_l2"], ac_f)
    # ---- (b) retrieval: bf16 query feature vs the reference's fp32 feature, planted 100k gallery, top-10
    q_ref = torch.from_numpy(g["feat"]).reshape(1, 256).float()
    gen = torch.Generator(device="cpu").manual_seed(77)
    rows = torch.nn.functional.normalize(torch.randn((100000, 256), generator=gen), dim=-1)
    where = 54321
    rows[where] = torch.nn.functional.normalize(q_ref + 0.1 * torch.randn((1, 256), generator=gen), dim=-1)[0]
    for j in range(12):                                            # a few near neighbours so that the top-10 is not all noise
        rows[1000 + 37 * j] = torch.nn.functional.normalize(q_ref + (0.3 + 0.05 * j) * torch.randn((1, 256), generator=gen), dim=-1)[0]
    rows = rows.to(BF16)
    rs, ri = oret.similarity_topk(q_ref, rows.float(), 10)                                   # fp32 feature, fp32 CPU product
    s_b, i_b = retrieval.GalleryShard(rows.to(DEV), 0).search(feat[:, 0].float(), 10)        # bf16-mode feature, HIP top-k
    mism = int((i_b.cpu() != ri).sum())
    rec1 = float((i_b[:, 0].cpu() == ri[:, 0]).float().mean())
    _note(name=f"full_depth_bf16_retrieval_{pooling}", recall_at_1_vs_fp32_oracle=rec1, planted_is_top1=bool(ri[0, 0] == where),
          top10_index_mismatches=mism, of=10, max_score_diff=float((s_b.cpu() - rs).abs().max()))
    assert ri[0, 0] == where and rec1 == 1.0
    # ---- (c) north_star's "bit-exact top-k indices", end to end in the mode that can deliver it: the exact-fp32
    # HIP forward's feature through cor_similarity_topk  vs  the REFERENCE's fp32 feature through the CPU chain top-k, same gallery
    # like with like: fp32 gallery holding the stored bf16 values (exact fp32 fmaf chain on the device) vs the chain oracle
    s_x, i_x = retrieval.GalleryShard(rows.float().to(DEV), 0).search(feat32[:, 0].float(), 10)
    rs, ri = oret.similarity_topk(q_ref, rows.float(), 10, exact_chain=True)
    mism32 = int((i_x.cpu() != ri).sum())
    _note(name=f"full_depth_fp32_mode_retrieval_{pooling}", fp32_mode_topk_index_mismatches_vs_reference=mism32, of=10,
          max_score_diff=float((s_x.cpu() - rs).abs().max()), feature_max_abs_err=float((feat32[:, 0].float().cpu() - q_ref).abs().max()))
    assert mism32 == 0, (i_x.cpu().tolist(), ri.tolist())


def test_attention_nonfinite_operands_stay_inside_their_head():
    """include/cor_amd.h: finite q / k / v are a PRECONDITION of the bf16 MFMA attention kernels (their softmax is compiled with
    -fno-honor-nans). What a violation may cost is bounded: a NaN / Inf in one (sample, head) leaves every OTHER
    (sample, head) bit-identical to the clean run - plain MHA (flash_fwd<0>), windowed SAM attention (win_attn: the poisoned window's
    head only) and global SAM attention (flash_global_pipe)."""
    ops, _ = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    B, H, T, hd = 2, 3, 256, 64
    d = H * hd
    qkv = torch.randn((B * T, 3 * d), generator=g, device=DEV).to(BF16)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    clean = ops.attention(q, k, v, B, H, T, T, hd, 0.125)
    bad = qkv.clone()
    bad[5, 1 * hd + 7] = float("nan")                    # q of sample 0, head 1
    bad[T + 9, d + 2 * hd + 3] = float("inf")            # k of sample 1, head 2
    out = ops.attention(bad[:, :d], bad[:, d:2 * d], bad[:, 2 * d:], B, H, T, T, hd, 0.125)
    keep = torch.ones((B, H), dtype=torch.bool)
    keep[0, 1] = keep[1, 2] = False
    for b in range(B):
        for h in range(H):
            if keep[b, h]:
                assert torch.equal(out[b * T:(b + 1) * T, h * hd:(h + 1) * hd], clean[b * T:(b + 1) * T, h * hd:(h + 1) * hd]), (b, h)
    # SAM attention, windowed (grid 28 = 2x2 windows) and global (grid 64)
    for grid, window in ((28, 14), (64, 0)):
        S = window or grid
        n = grid * grid
        x = torch.randn((B * n, 3 * d), generator=g, device=DEV).to(BF16)
        pad = torch.randn((3 * d,), generator=g, device=DEV).to(BF16)
        rh = torch.randn((2 * S - 1, hd), generator=g, device=DEV) * 0.3
        rw = torch.randn((2 * S - 1, hd), generator=g, device=DEV) * 0.3
        clean = ops.sam_attention(x, pad, rh, rw, B, H, grid, window)
        xb = x.clone()
        xb[n + 3 * grid + 2, 2 * d + 0 * hd + 11] = float("nan")     # v of sample 1, head 0, token (3, 2): window (0, 0) when windowed
        out = ops.sam_attention(xb, pad, rh, rw, B, H, grid, window)
        same = (out == clean).view(B, grid, grid, H, hd)
        assert bool(same[0].all()) and bool(same[1, :, :, 1:].all()), (grid, window)
        if window:
            assert bool(same[1, 14:].all()) and bool(same[1, :, 14:].all())                  # the other three windows of that head too


def test_batch32_bf16_vs_fp32_exact_mode_anchored_to_the_golden():
    """The benchmarked configuration's kernels (batch 32: persistent 256x256 GEMM, pipelined global attention, windowed block
    kernel, hipGraph-free eager launches) against the exact-fp32 HIP mode on the SAME 32 samples, with the fp32 mode itself
    pinned to the reference: sample 0 is the golden input of tests/golden/toplevel_MaskAdapterPooling.npz and its fp32 outputs
    must match the reference's (1e-3). Per-sample budgets for bf16 vs fp32 (the reference's own bf16-autocast error on this
    model is emb rel-L2 4.8e-2, masks 2.7e-2, 0.8 % sign flips: tests/golden/toplevel_autocast_bf16_*.npz):
    emb rel-L2 <= 3e-2, feat rel-L2 <= 1e-2, masks rel-L2 <= 6e-2, mask-sign flips <= 2 %, IoU-argmax flips reported."""
    from cor_amd import config
    g = load("toplevel_MaskAdapterPooling")
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    model = _build(12, (2, 5, 8, 11), gcfg, "MaskAdapterPooling")
    spec = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = ocfg.random_state({k: v for k, v in spec.items() if "attn_pool" not in k}, int(g["seed_params"]))
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).eval()
    B = 32
    gold = make_inputs(int(g["seed_inputs"]), q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512), mask=("mask", 1, 384))
    rest = make_inputs(4242, q=(B - 1, 3, 1024, 1024), s=(B - 1, 3, 384, 384), text=("tokens", B - 1, 64, 512), mask=("mask", B - 1, 384))
    kw = dict(query_image_inputs=torch.cat([gold["q"], rest["q"]]).to(DEV), support_image_inputs=torch.cat([gold["s"], rest["s"]]).to(DEV),
              change_text_inputs=torch.cat([gold["text"], rest["text"]]).to(DEV), support_mask_inputs=torch.cat([gold["mask"], rest["mask"]]).to(DEV))
    model.compute_dtype = F32
    parts = [model.forward_with_aux(**{k: v[i:i + 8] for k, v in kw.items()}, multimask_output=True) for i in range(0, B, 8)]
    m32, e32, f32_ = (torch.cat([p[j] for p in parts]) for j in range(3))
    best32 = torch.cat([p[3]["best"] for p in parts])
    es = float(np.abs(g["emb"]).max())
    report("batch32_fp32_sample0_emb_vs_reference_golden", e32[0:1, :, ::4, ::4], g["emb"], rtol=0, atol=1e-3 * es)
    report("batch32_fp32_sample0_masks_vs_reference_golden", m32[0:1, :, ::4, ::4], g["masks_1"], rtol=0, atol=1e-3 * float(np.abs(g["masks_1"]).max()))
    report("batch32_fp32_sample0_feat_vs_reference_golden", f32_[0:1], g["feat"], rtol=0, atol=1e-5)
    model.compute_dtype = BF16
    m16, e16, f16_, aux16 = model.forward_with_aux(**kw, multimask_output=True)
    rel = lambda a, b: ((a.float() - b.float()).flatten(1).norm(dim=1) / b.float().flatten(1).norm(dim=1))
    r_e, r_f, r_m = rel(e16, e32), rel(f16_, f32_), rel(m16, m32)
    flips = ((m16 > 0) != (m32 > 0)).flatten(1).float().mean(dim=1)
    argmax_flips = int((aux16["best"] != best32).sum())
    _note(name="batch32_bf16_vs_fp32_exact_mode", emb_rel_l2_max=float(r_e.max()), emb_rel_l2_mean=float(r_e.mean()), feat_rel_l2_max=float(r_f.max()),
          masks_rel_l2_max=float(r_m.max()), masks_rel_l2_mean=float(r_m.mean()), mask_sign_flip_fraction_max=float(flips.max()),
          mask_sign_flip_fraction_mean=float(flips.mean()), iou_argmax_flips=argmax_flips, of=B)
    assert float(r_e.max()) <= 3e-2 and float(r_f.max()) <= 1e-2, (r_e.max(), r_f.max())
    same = aux16["best"] == best32                                   # a flipped IoU argmax selects another mask channel: compare like with like
    assert float(r_m[same].max()) <= 6e-2 and float(flips[same].max()) <= 0.02, (r_m[same].max(), flips[same].max())
    assert argmax_flips <= 1, argmax_flips


def test_siglip_towers_vs_oracle():
    _, engine = _ops()
    from cor_amd import config
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    spec = ocfg.siglip_spec(gcfg, with_map_head=False)
    sd = ocfg.random_state(spec, 51)
    pk = packer(sd, F32)
    engine.pack_siglip(pk, gcfg)
    inp = make_inputs(52, s=(2, 3, 384, 384), text=("tokens", 2, 64, 512))
    vis = engine.siglip_vision(pk.W, inp["s"].to(DEV), gcfg, F32)
    report("siglip_vision_vs_oracle", vis.view(2, 576, 768), osig.vision_tokens(sd, inp["s"], gcfg), 1e-3, 1e-3)
    txt = engine.siglip_text(pk.W, inp["text"].to(DEV), gcfg, F32)
    report("siglip_text_vs_oracle", txt, osig.text_features(sd, inp["text"], gcfg), 1e-3, 1e-4)


# ======================================================================================================
# retrieval
# ======================================================================================================
# (Bq, Ng, k): small/ragged cases, then BASELINE configs[4]'s shard shape (512 x 125k) and the 1M-row single-GPU shard
TOPK_CASES = [(4, 1000, 5), (32, 10000, 10), (7, 37, 32), (64, 4097, 1), (300, 20011, 10), (512, 3000, 16), (64, 40000, 10), (300, 70001, 5),
              (64, 10000, 10), (257, 4096, 32), (512, 125000, 10), (512, 1000000, 10)]


@pytest.mark.parametrize("Bq,Ng,k", TOPK_CASES)
@pytest.mark.parametrize("gdt", [F32, BF16, torch.float16])
def test_similarity_topk(Bq, Ng, k, gdt):
    """north-star: top-k indices bit-identical to the CPU reference. EVERY gallery dtype is held BITWISE (scores and indices,
    ties included, zero excused positions) to the fmaf-chain oracle (oracle/c/sim_chain.c): fp32 galleries run the chain on
    the f32 MFMA; bf16 / fp16 galleries scan on the 16-bit MFMA and re-score the short list with the same chain over the
    stored values (query rounded to the gallery dtype)."""
    if gdt == F32 and Ng > 200000:
        pytest.skip("fp32 1M-row shard: 1 GB gallery through the exact-chain kernel; covered at 125k")
    ops, _ = _ops()
    rng = np.random.default_rng(Bq + Ng)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(gdt)
    G[5] = G[3]                                   # exact duplicate rows: ties must resolve to the smaller index
    if Ng > 300:
        G[Ng - 1] = G[17]; G[Ng - 200] = G[17]    # ... also across gallery slices and in the ragged last tile
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, g_offset=1000)
    Qr = Q if gdt == F32 else Q.to(gdt).float()
    rs, ri = oret.similarity_topk_chain(Qr, G.float(), k)
    kk = min(k, Ng)
    mism = int((i[:, :kk].cpu() - 1000 != ri).sum())
    bits = int((s[:, :kk].cpu().view(torch.int32) != rs.view(torch.int32)).sum())
    _note(name=f"topk_bitwise_{Bq}x{Ng}_k{k}_{gdt}", index_mismatches=mism, score_bit_mismatches=bits, entries=int(ri.numel()))
    assert mism == 0, f"{mism} of {ri.numel()} top-k indices differ from the chain oracle"
    assert bits == 0, f"{bits} of {ri.numel()} scores are not bit-identical to the chain oracle"
    if k > Ng:
        assert (i[:, Ng:] == -1).all() and torch.isinf(s[:, Ng:]).all()


@pytest.mark.parametrize("Bq,Ng,gdt,name", [(256, 100000, BF16, "configs[2]: 100k bf16 gallery, 8 shards of 12.5k, 8 x 32 queries"),
                                            (512, 1000000, torch.float16, "configs[4]: 1M fp16 gallery, 8 shards of 125k, 8 x 64 queries")])
def test_eight_way_sharded_search_equals_the_unsharded_oracle(Bq, Ng, gdt, name):
    """BASELINE configs[2] / configs[4] without the 8-GPU node: the gallery is cut into the 8 row shards shard_bounds() gives the 8
    ranks, every shard is searched on THIS GPU with its global offset by the same kernels a rank would run (all Bq all-gathered
    queries against its 12.5k / 125k rows), and the 8 packed lists go through the product's pack -> unpack -> merge_topk_host path.
    The merged top-10 must be BITWISE (scores and indices) the CPU chain oracle's over the whole gallery, duplicates across shard
    borders included. (What this cannot show is the RCCL transport itself: gloo world-2/-4 tests cover the collective logic.)"""
    from cor_amd import retrieval
    k = 10
    rng = np.random.default_rng(Bq + Ng)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(gdt)
    per = -(-Ng // 8)
    G[per] = G[per - 1]; G[5 * per + 3] = G[17]; G[Ng - 1] = G[17]                 # ties across shard borders / far-apart shards
    Gd = G.to(DEV)
    parts = []
    for r in range(8):
        lo, hi = retrieval.shard_bounds(Ng, 8, r)
        s_r, i_r = retrieval.GalleryShard(Gd[lo:hi], lo).search(Q.to(DEV), k)
        parts.append(retrieval._pack_lists(s_r, i_r).cpu())                          # what a rank sends in the gather
    ps, pi = retrieval._unpack_lists(torch.stack(parts))
    ms, mi = retrieval.merge_topk_host(list(ps), list(pi), k)
    rs, ri = oret.similarity_topk_chain(Q.to(gdt).float(), G.float(), k)
    mism = int((mi != ri).sum()); bits = int((ms.view(torch.int32) != rs.view(torch.int32)).sum())
    _note(name=f"eight_way_sharded_search_{Bq}x{Ng}_{gdt}", what=name, index_mismatches=mism, score_bit_mismatches=bits, entries=int(ri.numel()))
    assert mism == 0 and bits == 0, (mism, bits)


@pytest.mark.parametrize("Bq,Ng,k,gdt", [(512, 12500, 10, BF16), (256, 12500, 10, torch.float16), (32, 12500, 10, BF16), (8, 12500, 10, BF16),
                                         (32, 30000, 10, BF16), (64, 8193, 16, BF16), (300, 5000, 16, BF16), (300, 5000, 32, BF16), (33, 5000, 12, torch.float16),
                                         (1, 300, 5, BF16), (40, 255, 32, BF16), (512, 16384, 10, torch.float16)])
def test_similarity_small_shard_path_equals_the_global_threshold_path(Bq, Ng, k, gdt):
    """The two-launch local-threshold path of small shards (sim_block_scan + sim_final_wave: the 8-GPU shard shapes and few-query
    searches) against the five-launch global-threshold pipeline (COR_TOPK_FORCE_GLOBAL_THRESHOLD) and the CPU chain oracle: scores and
    indices BITWISE, duplicate rows across slice borders included; run twice (the candidate order inside the lists depends on LDS
    atomics, the answer must not)."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(Bq * 7 + Ng)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(gdt)
    if Ng > 1100:
        G[1024] = G[1023]; G[511] = G[512]; G[Ng - 1] = G[17]; G[Ng - 300] = G[17]
    G[5] = G[3]
    Qd, Gd = Q.to(DEV), G.to(DEV)
    s1, i1 = ops.similarity_topk(Qd, Gd, k, g_offset=7)
    s1b, i1b = ops.similarity_topk(Qd, Gd, k, g_offset=7)
    s2, i2 = ops.similarity_topk(Qd, Gd, k, g_offset=7, flags=nat.TOPK_FORCE_GLOBAL_THRESHOLD)
    # no query may take the brute-force fallback on a benign gallery: a sparse LAST slice (32 x 12 500: 212 rows) once left fewer
    # than k row classes populated, its threshold at -inf and every query on the fallback - exact, 25x slower
    _, raw = ops.similarity_topk(Qd, Gd, k, g_offset=7, flags=nat.TOPK_NO_FALLBACK)
    assert int((raw == -2).sum()) == 0
    assert torch.equal(i1, i1b) and torch.equal(s1.view(torch.int32), s1b.view(torch.int32))
    assert torch.equal(i1, i2) and torch.equal(s1.view(torch.int32), s2.view(torch.int32))
    rs, ri = oret.similarity_topk_chain(Q.to(gdt).float(), G.float(), k)
    kk = min(k, Ng)
    assert torch.equal(i1[:, :kk].cpu() - 7, ri) and torch.equal(s1[:, :kk].cpu().view(torch.int32), rs.view(torch.int32))
    _note(name=f"topk_small_path_{Bq}x{Ng}_k{k}_{gdt}", index_mismatches=0, score_bit_mismatches=0, entries=int(ri.numel()))


def test_similarity_small_shard_path_random_shapes():
    """Thirty seeded random (Bq, Ng, k, dtype) draws inside the small-shard path's domain (1..600 queries, 1..16 000 rows, k <= 16), ragged
    in every dimension: scores and indices bitwise equal to the CPU chain oracle, no fallback taken, duplicates at random places."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(2024)
    for case in range(30):
        Bq = int(rng.integers(1, 601)); Ng = int(rng.integers(1, 16001)); k = int(rng.integers(1, 17))
        gdt = BF16 if rng.integers(0, 2) else torch.float16
        Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
        G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(gdt)
        if Ng > 3:
            a, b = (int(v) for v in rng.integers(0, Ng, 2)); G[a] = G[b]
        s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, g_offset=3)
        _, raw = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, g_offset=3, flags=nat.TOPK_NO_FALLBACK)
        rs, ri = oret.similarity_topk_chain(Q.to(gdt).float(), G.float(), k)
        kk = min(k, Ng)
        assert int((raw == -2).sum()) == 0, (case, Bq, Ng, k)
        assert torch.equal(i[:, :kk].cpu() - 3, ri) and torch.equal(s[:, :kk].cpu().view(torch.int32), rs.view(torch.int32)), (case, Bq, Ng, k, gdt)
        if k > Ng:
            assert (i[:, Ng:] == -1).all()


@pytest.mark.parametrize("Bq,Ng,k,gdt", [(512, 125000, 10, torch.float16), (300, 70001, 5, BF16), (64, 40000, 10, BF16), (512, 3000, 16, BF16)])
def test_similarity_wave_selection_equals_the_block_selection_kernel(Bq, Ng, k, gdt):
    """Global-threshold pipeline, the two selection kernels against each other. Default (shipped): sim_prep -> sim_scan<SAMPLE> ->
    sim_scan<APPEND> (threshold ranked in its prologue from the 32 super-group maxima) -> block-per-query sim_final. Flag
    COR_TOPK_WAVE_FINAL: the same pipeline ending in sim_final_wave<RECORDS> (one wave per query; measured slower on this pipeline,
    kept as the A/B partner). Scores and indices bitwise equal, and both bitwise against the chain oracle."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(Bq * 3 + Ng)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(gdt)
    G[5] = G[3]; G[Ng - 1] = G[17]
    f = nat.TOPK_FORCE_GLOBAL_THRESHOLD
    s1, i1 = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, flags=f)
    s2, i2 = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, flags=f | nat.TOPK_WAVE_FINAL)
    assert torch.equal(i1, i2) and torch.equal(s1.view(torch.int32), s2.view(torch.int32))
    rs, ri = oret.similarity_topk_chain(Q.to(gdt).float(), G.float(), k)
    assert torch.equal(i1.cpu(), ri) and torch.equal(s1.cpu().view(torch.int32), rs.view(torch.int32))


@pytest.mark.parametrize("Bq,Ng,k", [(512, 125000, 32), (300, 70001, 24), (64, 40000, 17)])
def test_similarity_large_k_stays_off_the_fallback(Bq, Ng, k):
    """k > 16 on the global-threshold pipeline: tau is the k-th largest of 32 super-group maxima, a weak bound near k = 32;
    the record capacity is sized for it, so random data does not overflow a stream (COR_TOPK_NO_FALLBACK would expose -2), and the
    result is bitwise the chain oracle's."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(Bq + Ng + k)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1).to(BF16)
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), k, flags=nat.TOPK_NO_FALLBACK)
    assert int((i == -2).sum()) == 0, int((i == -2).any(dim=1).sum())
    rs, ri = oret.similarity_topk_chain(Q.to(BF16).float(), G.float(), k)
    assert torch.equal(i.cpu(), ri) and torch.equal(s.cpu().view(torch.int32), rs.view(torch.int32))


def test_similarity_small_shard_path_overflow_falls_back_on_the_device():
    """Degenerate small shard (every row identical): every slice list overflows; sim_final_wave flags the query and ranks the whole
    shard with the exact chain inside the same wave (COR_TOPK_NO_FALLBACK exposes the raw marker -2)."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(6)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((40, 256), dtype=np.float32)), dim=-1)
    row = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((1, 256), dtype=np.float32)), dim=-1)
    G = row.repeat(9000, 1).to(BF16)
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), 10)
    assert torch.equal(i.cpu(), torch.arange(10).repeat(40, 1)), i[:2]
    _, raw = ops.similarity_topk(Q.to(DEV), G.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK)
    assert (raw == -2).all()
    G2 = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((9000, 256), dtype=np.float32)), dim=-1)
    G2[2000:6000] = torch.nn.functional.normalize(Q[0:1] + 0.05 * row, dim=-1)
    G2 = G2.to(BF16)
    s2, i2 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10)
    _, raw2 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK)
    flagged = (raw2 == -2).all(dim=1).cpu()
    # LOCAL thresholds: a slice made of identical rows overflows its list for EVERY query (all its rows tie at the slice's own k-th
    # best), not only for the query they are close to. Round 5: an overflowing slice leaves its best score; sim_final_wave skips it when
    # that score is below the query's short-list cut, so only the query the duplicated row is close to takes the in-wave fallback
    # (rounds 1-4: every query did, ~25x the latency of the call)
    assert bool(flagged[0]) and int(flagged.sum()) <= 1, flagged
    _, raw3 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK | nat.TOPK_FORCE_GLOBAL_THRESHOLD)
    flagged3 = (raw3 == -2).all(dim=1).cpu()
    assert bool(flagged3[0]) and int(flagged3.sum()) < 40, flagged3
    rs, ri = oret.similarity_topk_chain(Q.to(BF16).float(), G2.float(), 10, margin=1e-3)
    assert torch.equal(i2.cpu(), ri) and torch.equal(s2.cpu().view(torch.int32), rs.view(torch.int32))
    # ... and every flagged query is one for which the duplicated row scores at (or just under: the cut is taken from a 16-bit key prefix
    # of the k-th best score among the OTHER slices' candidates, up to ~1 % low) its k-th best score: the duplicate slices cannot be
    # skipped for it; the other queries skipped them
    s_dup = Q.to(BF16).float() @ G2[2000].float()
    near = s_dup >= 0.95 * rs[:, 9] - 1e-3
    assert bool((near | ~flagged).all()), (flagged.nonzero().flatten(), near.nonzero().flatten(), s_dup[flagged], rs[flagged, 9])


def test_similarity_topk_lists_fallback_kernels():
    """The per-lane sorted-list kernels (COR_TOPK_FORCE_LISTS; what the device-side fallback runs after an overflow) rank by
    the 16-bit MFMA score: indices equal the chain oracle's except inside fp32-summation-order ties (counted)."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(99)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((300, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((20011, 256), dtype=np.float32)), dim=-1).to(BF16)
    G[5] = G[3]
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), 10, flags=nat.TOPK_FORCE_LISTS)
    rs, ri = oret.similarity_topk_chain(Q.to(BF16).float(), G.float(), 10)
    report("topk_lists_scores", s, rs, 1e-5, 2e-6)
    diff = i.cpu() != ri
    gap = torch.minimum(torch.cat([torch.ones(300, 1), rs[:, :-1] - rs[:, 1:]], 1), torch.cat([rs[:, :-1] - rs[:, 1:], torch.ones(300, 1)], 1))
    assert not (diff & (gap > 1e-5)).any()
    _note(name="topk_lists_fallback_kernels", positions_differing_inside_ties=int(diff.sum()), entries=int(ri.numel()))
    assert int(diff.sum()) <= 3


@pytest.mark.parametrize("Bq,Ng,k", [(8, 5000, 10), (33, 1237, 32)])
def test_similarity_topk_fp32_bitwise_vs_fma_chain_oracle(Bq, Ng, k):
    """fp32 gallery: the f32 MFMA is a k-ordered fmaf chain, restated in oracle/c/sim_chain.c -> scores must be
    BIT-IDENTICAL and hence the top-k indices identical, ties included (north-star: bit-exact top-k indices)."""
    ops, _ = _ops()
    rng = np.random.default_rng(Bq * Ng)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Bq, 256), dtype=np.float32)), dim=-1)
    G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((Ng, 256), dtype=np.float32)), dim=-1)
    G[11] = G[7]; G[Ng - 1] = G[2]
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), k)
    rs, ri = oret.similarity_topk(Q, G, k, exact_chain=True)
    assert torch.equal(i.cpu(), ri), "top-k indices differ from the bit-exact oracle"
    assert torch.equal(s.cpu().view(torch.int32), rs.view(torch.int32)), "scores are not bit-identical"


def test_similarity_topk_candidate_overflow_falls_back_on_the_device():
    """A degenerate gallery (every row identical => every score ties with the threshold) overflows the candidate buffers.
    sim_final detects it ON THE DEVICE and the same block ranks the whole shard for that query with the exact fmaf chain: the
    default call returns the exact answer with no host round trip and no second launch; COR_TOPK_NO_FALLBACK exposes the raw overflow marker (index -2). A mixed
    gallery (degenerate for some queries only) must repair exactly the flagged queries and leave the others bit-exact."""
    ops, _ = _ops()
    from cor_amd import _native as nat
    rng = np.random.default_rng(5)
    Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((40, 256), dtype=np.float32)), dim=-1)
    row = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((1, 256), dtype=np.float32)), dim=-1)
    G = row.repeat(50000, 1).to(BF16)
    s, i = ops.similarity_topk(Q.to(DEV), G.to(DEV), 10)
    assert torch.equal(i.cpu(), torch.arange(10).repeat(40, 1)), i[:2]
    _, raw = ops.similarity_topk(Q.to(DEV), G.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK)
    assert (raw == -2).all()
    # mixed: 30000 copies of a row close to query 0 only; the other queries see a benign gallery
    G2 = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((60000, 256), dtype=np.float32)), dim=-1)
    G2[10000:40000] = torch.nn.functional.normalize(Q[0:1] + 0.05 * row, dim=-1)
    G2 = G2.to(BF16)
    s2, i2 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10)
    _, raw2 = ops.similarity_topk(Q.to(DEV), G2.to(DEV), 10, flags=nat.TOPK_NO_FALLBACK)
    flagged = (raw2 == -2).all(dim=1).cpu()
    assert bool(flagged[0]) and int(flagged.sum()) < 40, flagged
    assert torch.equal(i2[0].cpu(), torch.arange(10000, 10010))
    rs, ri = oret.similarity_topk_chain(Q.to(BF16).float(), G2.float(), 10, margin=1e-3)
    # the in-kernel fallback ranks the flagged queries with the exact chain too: EVERY query is bit-identical to the oracle
    assert torch.equal(i2.cpu(), ri) and torch.equal(s2.cpu().view(torch.int32), rs.view(torch.int32))


# ======================================================================================================
# inference harness: post-processing, metrics, save_hard_pred_masks / val_metric
# ======================================================================================================
def test_postprocess_resize_binarize_and_metrics_vs_oracle():
    ops, _ = _ops()
    rng = np.random.default_rng(8)
    logits = torch.from_numpy(rng.standard_normal((3, 1, 256, 256), dtype=np.float32) * 3)
    _, soft = oret.postprocess_masks(logits)
    prob = ops.mask_prob_minmax(logits.to(DEV))
    report("harness_prob_minmax", prob, soft, 1e-5, 1e-6)
    for oh, ow in ((480, 640), (256, 256), (100, 77)):
        hard = ops.resize_binarize(prob, oh, ow, 0.5)
        ref_hard, _ = oret.postprocess_masks(logits, out_hw=(oh, ow))
        diff = (hard.cpu() != ref_hard[:, 0]).float().mean().item()
        assert diff <= 2e-5, f"binarised masks differ on {diff:.2e} of the pixels at {(oh, ow)}"   # only exact-0.5 crossings
    gt = torch.from_numpy((rng.random((3, 1, 256, 256)) > 0.6).astype(np.float32))
    report("harness_metrics", ops.mask_metrics(prob, gt.to(DEV)), oret.mask_metrics(soft, gt), 1e-4, 1e-6)


def test_harness_save_hard_pred_masks_and_val_metric(tmp_path):
    """End-to-end harness loop on the GPU with a stub model that returns fixed logits: PNG files (name, size, content)
    and the metrics CSV must match the oracle's post-processing."""
    import logging
    from types import SimpleNamespace
    from PIL import Image
    from cor_amd import harness
    rng = np.random.default_rng(9)
    logits = torch.from_numpy(rng.standard_normal((2, 1, 256, 256), dtype=np.float32) * 2)

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
        def forward(self, **kw):
            return logits.to(self.w.device), None, None

    root = tmp_path / "data"
    (root / "dsA" / "mask" / "7").mkdir(parents=True)
    sizes = [(320, 200), (123, 456)]                     # PIL size = (W, H)
    gts = []
    for i, (w, h) in enumerate(sizes):
        g = (rng.random((h, w)) > 0.5).astype(np.uint8) * 255
        Image.fromarray(g).save(root / "dsA" / "mask" / "7" / f"m{i}.png")
        gts.append(g)
    batch = dict(query_img=torch.zeros(2, 3, 8, 8), support_img=torch.zeros(2, 3, 8, 8), support_mask=torch.zeros(2, 1, 8, 8),
                 text=torch.ones(2, 64, dtype=torch.long), pair_id=[11, 12], query_mask_name=["m0.png", "m1.png"],
                 dataset=["dsA", "dsA"], target=[7, 7], query_mask=(torch.from_numpy(rng.random((2, 1, 256, 256))) > 0.5).float(),
                 compose=[0, 1], query_cat=[3, 4])
    opt = SimpleNamespace(vaild_model_save_path=str(tmp_path / "out"), multimask_output=True)
    log = logging.getLogger("harness-test")
    model = Stub().to(DEV)
    harness.save_hard_pred_masks([batch], model, opt, log, None, dataset_path=str(root), pred_save_dir="pred")
    for i, (w, h) in enumerate(sizes):
        got = np.array(Image.open(tmp_path / "out" / "pred" / f"{11 + i}_m{i}.png"))
        ref, _ = oret.postprocess_masks(logits[i:i + 1], out_hw=(h, w))
        assert got.shape == (h, w) and got.dtype == np.uint8
        assert (got != ref[0, 0].numpy()).mean() <= 2e-5
    # soft masks (utils/vailder.py:513-656): the resized probabilities as (p * 255) truncated to uint8
    harness.save_soft_pred_masks([batch], model, opt, log, None, dataset_path=str(root), pred_save_dir="soft")
    for i, (w, h) in enumerate(sizes):
        got = np.array(Image.open(tmp_path / "out" / "soft" / f"{11 + i}_m{i}.png")).astype(np.int32)
        _, prob = oret.postprocess_masks(logits[i:i + 1], out_hw=(h, w))
        ref = (prob[0, 0] * 255).to(torch.uint8).numpy().astype(np.int32)
        assert got.shape == (h, w)
        assert np.abs(got - ref).max() <= 1 and (got != ref).mean() <= 2e-3       # truncation next to an integer boundary
    res = harness.val_metric([batch], model, opt, log, None)
    _, soft = oret.postprocess_masks(logits)
    ref_m = oret.mask_metrics(soft, batch["query_mask"]).mean(0)
    got_m = res["global_metrics"]
    for k, j in (("dice", 0), ("mae", 1), ("iou", 2), ("mdice", 3), ("miou", 4)):
        assert abs(got_m[k] - ref_m[j].item()) < 1e-4, (k, got_m[k], ref_m[j].item())
    rows = list(__import__("csv").DictReader(open(tmp_path / "out" / "per_sample_metrics.csv")))
    assert len(rows) == 2 and rows[0]["Id"] == "11" and rows[1]["Query_mask"] == "m1.png"


def test_my_test_driver_end_to_end_real_model(tmp_path):
    """cor_amd.my_test (the reference's my_test.py:49-234) end to end on a synthetic dataset with the REAL model (full SAM-B +
    SigLIP-B/16, random weights): YAML -> factory -> CSV loaders (Compose != 0 rows dropped, GPU resize in the collate step)
    -> strict checkpoint load ("model_state_dict" + "module." prefix) -> hard / soft PNGs at the ground-truth size + the
    per-sample metrics CSV. Run twice: `mixed_precision: "no"` (exact fp32 mode) against the CPU oracle on the same files
    (Pillow resize restated bit-exactly, fp32 forward, post-processing): >= 99.9 % of the hard pixels equal, soft masks within
    2 grey levels; `bf16` against the same HIP model called directly (plumbing check; the bf16 parity budgets live in the
    full-depth tests: with random weights the min-max normalised threshold amplifies the bf16 error).)"""
    import csv as _csv
    import yaml
    from PIL import Image
    from cor_amd import my_test, tokenizer, utils
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    from oracle import preprocess as OP
    rng = np.random.default_rng(21)
    root = tmp_path / "data"
    for d in ("image", "mask/cat", "mask/sup"):
        (root / "dsA" / d).mkdir(parents=True)
    rows = []
    for i, (w, h) in enumerate([(333, 250), (200, 301), (256, 256)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "dsA" / "image" / f"q{i}.png")
        Image.fromarray(rng.integers(0, 256, (h - 7, w + 5, 3), dtype=np.uint8)).save(root / "dsA" / "image" / f"s{i}.png")
        m = np.zeros((h, w), np.uint8); m[h // 4: h // 2, w // 3: w // 3 * 2] = 255
        Image.fromarray(m).save(root / "dsA" / "mask" / "cat" / f"qm{i}.png")
        sm = np.zeros((h - 7, w + 5), np.uint8); sm[10:90, 20:150] = 255
        Image.fromarray(sm).save(root / "dsA" / "mask" / "sup" / f"sm{i}.png")
        rows.append(dict(Id=100 + i, Query_img=f"q{i}.png", Query_mask=f"qm{i}.png", Support_img=f"s{i}.png", Support_mask=f"sm{i}.png",
                         Text=f"make the cat number {i} larger, please!", Compose=0, Dataset="dsA", Target="cat", query_cat=3))
    dropped = dict(rows[0], Id=999, Compose=1)                                      # Compose != 0: the loader must drop it
    cols = ["Id", "Query_img", "Query_mask", "Support_img", "Support_mask", "Text", "Compose", "Dataset", "Target", "query_cat"]
    for name, rr in (("Test_1.csv", [rows[0], dropped, rows[1]]), ("Test_2.csv", [rows[2]])):
        with open(tmp_path / name, "w", newline="") as f:
            wr = _csv.DictWriter(f, fieldnames=cols); wr.writeheader(); wr.writerows(rr)
    donor = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    utils.randomize_parameters(donor, seed=13)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    torch.save({"model_state_dict": {f"module.{k}": v for k, v in sd.items()}, "epoch": 3}, tmp_path / "ckpt.pth")
    cfg = dict(batch_size=2, sam_model_name="sam_base", siglip_model_name="ViT-B-16-SigLIP-384", dataset_path=str(root), val_csv_A=str(tmp_path / "Test_1.csv"),
               val_csv_B=str(tmp_path / "Test_2.csv"), vaild_model_save_path=str(tmp_path / "out"), mask_pooling="MaskAdapterPooling", multimask_output=False,
               load_checkpoint_path=str(tmp_path / "ckpt.pth"), num_workers=0)
    tok = tokenizer.hashing_tokenizer(vocab=32000)
    def prep(path, size, norm):
        a = np.asarray(Image.open(path).convert("RGB" if norm else "L"))
        u8 = OP.resize_bilinear_u8(a, size, size)
        return torch.from_numpy(OP.to_tensor_normalize(u8, OP.IMAGENET_MEAN if norm else None, OP.IMAGENET_STD if norm else None))
    q = torch.stack([prep(root / "dsA" / "image" / f"q{i}.png", 1024, True) for i in range(3)])
    s_ = torch.stack([prep(root / "dsA" / "image" / f"s{i}.png", 384, True) for i in range(3)])
    sm_ = torch.stack([prep(root / "dsA" / "mask" / "sup" / f"sm{i}.png", 384, False) for i in range(3)])
    txt = torch.stack([tok(r["Text"]) for r in rows])

    def run(mp, out, accelerate=False):
        c = dict(cfg, mixed_precision=mp, vaild_model_save_path=str(tmp_path / out))
        with open(tmp_path / f"cfg_{out}.yaml", "w") as f:
            yaml.safe_dump(c, f)
        my_test.main(["--config", str(tmp_path / f"cfg_{out}.yaml"), "--soft", "1", "--metric", "1"] + (["--accelerate", "1"] if accelerate else []))

    def pngs(out, i, d):
        return (np.array(Image.open(tmp_path / out / f"hard_pred_{d}" / f"{100 + i}_qm{i}.png")),
                np.array(Image.open(tmp_path / out / f"soft_pred_{d}" / f"{100 + i}_qm{i}.png")).astype(np.int32))

    model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()

    def direct(autocast):
        ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else __import__("contextlib").nullcontext()
        with ctx:
            return torch.cat([model(query_image_inputs=q[j:j + 2].to(DEV), support_image_inputs=s_[j:j + 2].to(DEV), change_text_inputs=txt[j:j + 2].to(DEV),
                                    support_mask_inputs=sm_[j:j + 2].to(DEV), multimask_output=False)[0] for j in (0, 2)]).float().cpu()

    ref_masks, _, _ = omodel.forward(sd, "sam_base", "ViT-B-16-SigLIP-384", "MaskAdapterPooling", q, s_, txt, sm_, False)
    for mp, out in (("no", "out32"), ("bf16", "out16"), ("bf16", "out16acc")):
        run(mp, out, accelerate=out.endswith("acc"))                 # third run: Accelerator(bf16) -> prepare(model) -> accelerator.autocast()
        logits = direct(mp == "bf16")
        assert not (tmp_path / out / "hard_pred_Test_1" / "999_qm0.png").exists()
        for i, d in ((0, "Test_1"), (1, "Test_1"), (2, "Test_2")):
            hard, soft = pngs(out, i, d)
            gt = np.array(Image.open(root / "dsA" / "mask" / "cat" / f"qm{i}.png"))
            assert hard.shape == gt.shape and set(np.unique(hard)) <= {0, 255}
            # (a) plumbing: the driver's files against the same HIP model called directly on the oracle-preprocessed tensors
            # (CSV, decode, bit-exact GPU resize in the collate step, checkpoint, naming, sizes)
            rh, rp = oret.postprocess_masks(logits[i:i + 1], out_hw=gt.shape)
            agree = float((hard == rh[0, 0].numpy()).mean())
            dsoft = np.abs(soft - (rp[0, 0] * 255).to(torch.uint8).numpy().astype(np.int32))
            _note(name=f"my_test_driver_{mp}_sample{i}_vs_direct_call", hard_pixel_agreement=agree, soft_max_grey_diff=int(dsoft.max()))
            assert agree >= 0.9995 and dsoft.max() <= 1, (mp, i, agree, dsoft.max())
            if out == "out16acc":                                     # the accelerate path writes the SAME files as the plain bf16 run, bit for bit
                h2, s2 = pngs("out16", i, d)
                assert np.array_equal(hard, h2) and np.array_equal(soft, s2), (i, d)
            if mp == "no":
                # (b) exact fp32 mode against the fp32 CPU oracle run on the same files. Budget: with these random weights the mask
                # decoder is ill-conditioned (test_fp32_mode_per_stage_vs_fp64_oracle measures it on these very inputs: a logit
                # moves 2.7e4 .. 3.4e4 per unit of embedding error, torch fp32 itself ends 0.66 of 5.1 away from an fp64
                # evaluation, the HIP fp32 mode 0.94 end to end and 0.08 - against torch's 0.38 - with exact decoder inputs), so
                # two fp32 evaluations differ by up to ~1 logit at isolated pixels; the tight fp32 bounds live in the golden tests
                oh, op_ = oret.postprocess_masks(ref_masks[i:i + 1], out_hw=gt.shape)
                agree = float((hard == oh[0, 0].numpy()).mean())
                dsoft = np.abs(soft - (op_[0, 0] * 255).to(torch.uint8).numpy().astype(np.int32))
                dl = float((logits[i] - ref_masks[i]).abs().max())
                _note(name=f"my_test_driver_fp32_sample{i}_vs_oracle", hard_pixel_agreement=agree, soft_max_grey_diff=int(dsoft.max()), soft_mean_grey_diff=float(dsoft.mean()),
                      logits_max_abs_diff=dl, logits_scale=float(ref_masks[i].abs().max()))
                assert agree >= 0.995 and dsoft.mean() <= 1.0, (i, agree, dsoft.mean())
    got = list(_csv.DictReader(open(tmp_path / "out32" / "per_sample_metrics_Test_1.csv")))
    assert [r["Id"] for r in got] == ["100", "101"] and got[0]["Text"].startswith("make the cat")


def test_accelerator_prepare_and_autocast_equal_the_direct_bf16_call():
    """the module must survive accelerator.prepare() (my_test.py:108) and run under accelerator.autocast()
    (utils/vailder.py:416; config/vaild_config/vaild_a.yaml:4 mixed_precision bf16). The prepared model under the accelerator's
    autocast must give bit for bit what the direct call under torch.autocast(bf16) gives; state_dict keys survive prepare()."""
    from accelerate import Accelerator
    from cor_amd import config
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    model = _build(2, (1,), gcfg, "MaskAdapterPooling")
    sd = ocfg.random_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, 91)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    inp = make_inputs(92, q=(2, 3, 1024, 1024), s=(2, 3, 384, 384), text=("tokens", 2, 64, 512), mask=("mask", 2, 384))
    kw = dict(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV), change_text_inputs=inp["text"].to(DEV),
              support_mask_inputs=inp["mask"].to(DEV), multimask_output=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        want = model(**kw)
    acc = Accelerator(mixed_precision="bf16")
    prepared = acc.prepare(model)
    assert set(acc.unwrap_model(prepared).state_dict().keys()) == set(sd.keys())
    prepared.eval()
    with torch.no_grad(), acc.autocast():
        got = prepared(**kw)
    for a, b, name in zip(got, want, ("masks", "emb", "feat")):
        assert a.dtype == torch.float32 and torch.equal(a, b), name
    with torch.no_grad():                                          # outside autocast the prepared model runs its own compute_dtype (fp32 exact)
        exact = prepared(**kw)
    direct = model(**kw)
    assert all(torch.equal(a.float(), b.float()) for a, b in zip(exact, direct))


def _driver_like_inputs(n=2):
    """The inputs test_my_test_driver_end_to_end_real_model feeds (same generator order, no files): uint8 images of odd sizes
    through the oracle's Pillow-exact resize + normalisation, the rectangle support masks, hashed tokens."""
    from cor_amd import tokenizer
    from oracle import preprocess as OP
    rng = np.random.default_rng(21)
    tok = tokenizer.hashing_tokenizer(vocab=32000)
    q, s_, sm_, txt = [], [], [], []
    for i, (w, h) in enumerate([(333, 250), (200, 301), (256, 256)][:n]):
        qi = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        si = rng.integers(0, 256, (h - 7, w + 5, 3), dtype=np.uint8)
        sm = np.zeros((h - 7, w + 5), np.uint8); sm[10:90, 20:150] = 255
        q.append(torch.from_numpy(OP.to_tensor_normalize(OP.resize_bilinear_u8(qi, 1024, 1024), OP.IMAGENET_MEAN, OP.IMAGENET_STD)))
        s_.append(torch.from_numpy(OP.to_tensor_normalize(OP.resize_bilinear_u8(si, 384, 384), OP.IMAGENET_MEAN, OP.IMAGENET_STD)))
        sm_.append(torch.from_numpy(OP.to_tensor_normalize(OP.resize_bilinear_u8(sm, 384, 384), None, None)))
        txt.append(tok(f"make the cat number {i} larger, please!"))
    return torch.stack(q), torch.stack(s_), torch.stack(sm_), torch.stack(txt)


def test_fp32_mode_per_stage_vs_fp64_oracle():
    """Where does the exact-fp32 HIP mode stand against an fp64 evaluation of the reference formula, stage by stage, on the
    driver test's uint8-derived inputs and batch of 2 (round 2 saw |logit| differences up to 0.76 of 4.7 against the fp32 oracle
    there, against 2e-5 on the N(0,1) goldens)? Every stage is measured twice: HIP fp32 vs oracle fp64, and oracle fp32 (torch
    CPU) vs oracle fp64 - the second column is what fp32 arithmetic itself costs at that stage. The decoder stages are run
    ISOLATED (both fed the fp64 oracle's embedding and feature rounded to fp32), so an amplifying stage shows up as such and
    not as inherited error. Assertions: HIP fp32 is within 4x of torch fp32's own distance to fp64 at every isolated stage and at
    the two encoder outputs, and within 8x at every END-TO-END decoder stage (plus a floor of 2e-6 of the stage's scale), so a
    regression of the encoder's fp32 path cannot hide behind the decoder's conditioning; the table goes to the parity report."""
    from cor_amd import utils, engine
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=13)
    sd32 = {k: v.clone() for k, v in model.state_dict().items()}
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd32.items()}
    q, s_, sm_, txt = _driver_like_inputs(2)
    B = q.shape[0]
    model = model.to(DEV).eval()
    model.compute_dtype = F32
    W = model.packed(F32)
    scfg, gcfg = model.image_encoder.cfg, model.support_branch.siglip.cfg

    def oracle(sd, dt, emb_in=None, feat_in=None):
        out = {}
        if emb_in is None:
            out["sam_embedding"] = osam.image_encoder(sd, q.to(dt), scfg)
            out["support_feature"] = osup.support_branch(sd, s_.to(dt), txt, sm_.to(dt), gcfg, "MaskAdapterPooling")
            emb_in, feat_in = out["sam_embedding"], out["support_feature"]
        tr = {}
        masks, iou, _ = osam.mask_decoder(sd, emb_in.to(dt), osam.dense_pe(sd), feat_in.to(dt), osam.dense_no_mask(sd, B), False, trace=tr)
        out.update({k: v for k, v in tr.items() if v is not None})
        return out

    o64 = oracle(sd64, torch.float64)
    emb64_as32, feat64_as32 = o64["sam_embedding"].float(), o64["support_feature"].float()
    o32_full = oracle(sd32, F32)                                           # end to end in fp32
    o32_iso = oracle(sd32, F32, emb64_as32, feat64_as32)                   # decoder alone, exact inputs
    o64_iso = oracle(sd64, torch.float64, emb64_as32, feat64_as32)         # the same rounded inputs in fp64: the isolated stages' truth

    # HIP fp32: the same stages through the C ABI
    with torch.cuda.device(torch.device(DEV)):
        emb_tok = engine.sam_encoder(W, q.to(DEV), scfg, F32)
        vis = engine.siglip_vision(W, s_.to(DEV), gcfg, F32)
        tx = engine.siglip_text(W, txt.to(DEV), gcfg, F32)
        feat = engine.support_head(W, vis, tx, sm_.to(DEV), gcfg, "MaskAdapterPooling", F32)
        ops, _ = _ops()
        g = 64
        hip_full = {"sam_embedding": ops.tokens_to_nchw(emb_tok, B, g * g, 256).view(B, 256, g, g), "support_feature": feat.view(B, 1, -1)}
        tr = {}
        engine.mask_decoder(W, emb_tok, feat, F32, False, all_masks=True, trace=tr)
        hip_full.update(tr)
        tr = {}
        emb_iso = ops.nchw_to_tokens(emb64_as32.to(DEV).contiguous(), F32)
        engine.mask_decoder(W, emb_iso, feat64_as32.to(DEV).view(B, -1).contiguous(), F32, False, all_masks=True, trace=tr)
        hip_iso = tr
    torch.cuda.synchronize()

    def dist(a, b):
        a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
        assert a.numel() == b.numel(), (a.shape, b.shape)
        return float((a - b).abs().max()), float(b.abs().max())

    stages = ["sam_embedding", "support_feature", "tokens_l0", "keys_l0", "tokens_l1", "keys_l1", "hs", "upscaled1", "hyper", "iou", "masks_all"]
    worst, worst_e2e = [], []
    for st in stages:
        for kind, hip, o32, truth in (("end_to_end", hip_full, o32_full, o64), ("isolated", hip_iso, o32_iso, o64_iso)):
            if st not in hip or st not in truth:
                continue
            eh, scale = dist(hip[st], truth[st])
            eo, _ = dist(o32[st], truth[st])
            rec = dict(name=f"fp32_stage_table_{kind}_{st}", hip32_vs_oracle64=eh, oracle32_vs_oracle64=eo, scale=scale, ratio=eh / max(eo, 1e-300))
            _note(**rec)
            if kind == "isolated" or st in ("sam_embedding", "support_feature"):
                if eh > 4.0 * eo + 2e-6 * scale:
                    worst.append(rec)
            elif eh > 8.0 * eo + 2e-6 * scale:
                # end-to-end decoder stages inherit the encoder's error times the decoder's gain (3e4 per unit with this init): the
                # conditioning argument is a BOUND here, not a comment - measured worst 6.7x (tokens_l0), 5.9x (iou)
                worst_e2e.append(rec)
    # the end-to-end logits: inherited embedding error times the decoder's own amplification (reported, bounded loosely)
    eh, scale = dist(hip_full["masks_all"], o64["masks_all"])
    eo, _ = dist(o32_full["masks_all"], o64["masks_all"])
    amp_h = eh / max(dist(hip_full["sam_embedding"], o64["sam_embedding"])[0], 1e-30)
    amp_o = eo / max(dist(o32_full["sam_embedding"], o64["sam_embedding"])[0], 1e-30)
    _note(name="fp32_stage_table_amplification", hip_logit_err=eh, oracle32_logit_err=eo, logit_scale=scale,
          hip_logit_err_per_unit_embedding_err=amp_h, oracle32_logit_err_per_unit_embedding_err=amp_o)
    assert not worst, f"HIP fp32 is more than 4x further from fp64 than torch fp32 at: {worst}"
    assert not worst_e2e, f"end to end, HIP fp32 is more than 8x further from fp64 than torch fp32 at: {worst_e2e}"


def test_gallery_builder_and_checkpoint_loader(tmp_path):
    """CORE-style checkpoint ("model_state_dict", "module." prefix) loads strictly; the gallery
    builder's rows equal loss_func.mask_pooling of the oracle's SAM embeddings; shards round-trip through disk."""
    from cor_amd import config, harness, retrieval
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=1, t_depth=1, vocab=64)
    model = _build(2, (1,), gcfg, "MaskedPooling")
    sd = ocfg.random_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, 61)
    torch.save({"epoch": 3, "model_state_dict": {"module." + k: v for k, v in sd.items()}}, tmp_path / "ck.pth")
    res, epoch = harness.load_core_checkpoint(model, str(tmp_path / "ck.pth"))
    assert epoch == 3 and not res.missing_keys and not res.unexpected_keys
    model = model.to(DEV).eval()
    inp = make_inputs(62, q=(2, 3, 1024, 1024), m=("mask", 2, 256))
    rows = retrieval.build_gallery(model, [dict(query_img=inp["q"], query_mask=inp["m"])], dtype=torch.float32)
    ref = oret.region_embedding(osam.image_encoder(sd, inp["q"], dict(model.image_encoder.cfg)), inp["m"])[:, 0]
    report("gallery_rows_vs_oracle", rows, ref, 1e-3, 1e-4)
    retrieval.save_gallery(str(tmp_path / "gal"), rows.to(torch.float16), world=2)
    sh = retrieval.load_gallery_shard(str(tmp_path / "gal"), 1, DEV)
    assert sh.offset == 1 and len(sh) == 1 and torch.equal(sh.rows.cpu(), rows[1:2].to(torch.float16).cpu())
    # the same builder fed from the reference's CSV schema (utils/dataloader.py:244-369): files decoded on the host, Pillow-exact
    # resize on the GPU (cor_amd.dataloader.gallery_batches); rows with Compose != 0 are dropped
    import csv as _csv
    from PIL import Image
    from cor_amd import dataloader
    from oracle import preprocess as OP
    rng = np.random.default_rng(63)
    root = tmp_path / "data"
    (root / "ds" / "image").mkdir(parents=True); (root / "ds" / "mask" / "dog").mkdir(parents=True)
    recs = []
    for i, (w, h) in enumerate([(300, 220), (180, 240), (256, 256)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "ds" / "image" / f"q{i}.png")
        mk = np.zeros((h, w), np.uint8); mk[h // 5: h // 2, w // 4: w // 2] = 255
        Image.fromarray(mk).save(root / "ds" / "mask" / "dog" / f"m{i}.png")
        recs.append(dict(Id=i, Query_img=f"q{i}.png", Query_mask=f"m{i}.png", Support_img="x.png", Support_mask="x.png", Text="t", Compose=int(i == 1),
                         Dataset="ds", Target="dog", query_cat=0))
    with open(tmp_path / "gal.csv", "w", newline="") as f:
        wr = _csv.DictWriter(f, fieldnames=dataloader.CSV_COLUMNS); wr.writeheader(); wr.writerows(recs)
    rows2 = retrieval.build_gallery(model, dataloader.gallery_batches(str(tmp_path / "gal.csv"), str(root), batch_size=2, device=DEV), dtype=torch.float32)
    assert rows2.shape == (2, 256)                                                 # the Compose == 1 row is not a gallery entry
    keep = [0, 2]
    qi = torch.stack([torch.from_numpy(OP.to_tensor_normalize(OP.resize_bilinear_u8(np.asarray(Image.open(root / "ds" / "image" / f"q{i}.png").convert("RGB")), 1024, 1024),
                                                              OP.IMAGENET_MEAN, OP.IMAGENET_STD)) for i in keep])
    mi = torch.stack([torch.from_numpy(OP.to_tensor_normalize(OP.resize_bilinear_u8(np.asarray(Image.open(root / "ds" / "mask" / "dog" / f"m{i}.png").convert("L")), 1024, 1024),
                                                              None, None)) for i in keep])
    ref2 = oret.region_embedding(osam.image_encoder(sd, qi, dict(model.image_encoder.cfg)), mi)[:, 0]
    report("gallery_rows_from_csv_vs_oracle", rows2, ref2, 1e-3, 1e-4)


# ======================================================================================================
# other accepted model names and edge inputs
# ======================================================================================================
def _build_any(sam_dim, sam_heads, siglip_name, gcfg, pooling, depth=2, gidx=(1,)):
    from cor_amd.lib.sam_model.image_encoder import ImageEncoderViT
    from cor_amd.lib.sam_model.mask_decoder import MaskDecoder
    from cor_amd.lib.sam_model.my_prompt_encoder import PromptEncoder
    from cor_amd.lib.sam_model.transformer import TwoWayTransformer
    from cor_amd.lib.sam_with_sup_branch import CirSegModelWithQuerySupportFeat
    from cor_amd.lib.support_branch import SupportBranch
    return CirSegModelWithQuerySupportFeat(
        image_encoder=ImageEncoderViT(embed_dim=sam_dim, depth=depth, num_heads=sam_heads, global_attn_indexes=gidx),
        support_branch=SupportBranch(siglip_name, None, pooling, siglip_cfg=gcfg),
        prompt_encoder=PromptEncoder(256, (64, 64)),
        mask_decoder=MaskDecoder(transformer_dim=256, transformer=TwoWayTransformer(2, 256, 8, 2048)))


@pytest.mark.parametrize("sam_dim,sam_heads,siglip_name", [(1280, 16, "ViT-SO400M-14-SigLIP-384"), (1024, 16, "ViT-L-16-SigLIP2-384")])
def test_other_model_sizes_vs_oracle(sam_dim, sam_heads, siglip_name):
    """SAM-H width (head_dim 80) + SO400M/14 (27x27 tokens, head_dim 72, K = 588 / 4304: ragged GEMM K) and
    SAM-L width + SigLIP2-L (tanh GELU), depth-reduced; fp32 exact mode against the oracle, bf16 mode sanity."""
    from cor_amd import config
    gcfg = dict(config.siglip_cfg(siglip_name), depth=1, t_depth=1, vocab=128)
    model = _build_any(sam_dim, sam_heads, siglip_name, gcfg, "MaskAdapterPooling")
    sd = ocfg.random_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, 71)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    inp = make_inputs(72, q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 128), mask=("mask", 1, 384))
    scfg = dict(model.image_encoder.cfg)
    ref_emb = osam.image_encoder(sd, inp["q"], scfg)
    ref_feat = osup.support_branch(sd, inp["s"], inp["text"], inp["mask"], gcfg, "MaskAdapterPooling")
    ref_masks, ref_iou, _ = osam.mask_decoder(sd, ref_emb, osam.dense_pe(sd), ref_feat, osam.dense_no_mask(sd, 1), True)
    kw = dict(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV), change_text_inputs=inp["text"].to(DEV),
              support_mask_inputs=inp["mask"].to(DEV))
    masks, emb, feat, aux = model.forward_with_aux(**kw, multimask_output=True)
    report(f"sizes_{sam_dim}_{siglip_name}_emb", emb, ref_emb, 1e-3, 2e-3)
    report(f"sizes_{sam_dim}_{siglip_name}_feat", feat, ref_feat, 1e-3, 1e-4)
    report(f"sizes_{sam_dim}_{siglip_name}_masks", aux["masks"][:, 1:], ref_masks, 2e-3, 5e-3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, emb_b, feat_b = model(**kw, multimask_output=True)
    report(f"sizes_{sam_dim}_{siglip_name}_emb_bf16", emb_b, ref_emb, 0, 4e-2 * ref_emb.abs().max().item())
    report(f"sizes_{sam_dim}_{siglip_name}_feat_bf16", feat_b, ref_feat, 0, 3e-2)


def test_edge_inputs_vs_oracle():
    """B=3 (not a tile multiple), an all-zero support mask, an all-one mask, an all-pad text; MaskedPooling divides by
    (sum + 1e-8) like the reference (mask_adapter.py:22)."""
    from cor_amd import config
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=1, t_depth=1, vocab=64)
    for pooling in ("MaskedPooling", "MaskAdapterPooling"):
        model = _build(2, (1,), gcfg, pooling)
        sd = ocfg.random_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, 81)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).eval()
        inp = make_inputs(82, q=(3, 3, 1024, 1024), s=(3, 3, 384, 384), text=("tokens", 3, 64, 64), mask=("mask", 3, 384))
        inp["mask"][0] = 0.0
        inp["mask"][1] = 1.0
        inp["text"][2] = 1
        scfg = dict(model.image_encoder.cfg)
        ref_emb = osam.image_encoder(sd, inp["q"], scfg)
        ref_feat = osup.support_branch(sd, inp["s"], inp["text"], inp["mask"], gcfg, pooling)
        ref_masks, ref_iou, _ = osam.mask_decoder(sd, ref_emb, osam.dense_pe(sd), ref_feat, osam.dense_no_mask(sd, 3), False)
        masks, emb, feat = model(query_image_inputs=inp["q"].to(DEV), support_image_inputs=inp["s"].to(DEV),
                                 change_text_inputs=inp["text"].to(DEV), support_mask_inputs=inp["mask"].to(DEV), multimask_output=False)
        report(f"edge_{pooling}_feat", feat, ref_feat, 1e-3, 1e-4)
        report(f"edge_{pooling}_masks", masks, ref_masks, 2e-3, 5e-3)
        assert torch.isfinite(masks).all() and torch.isfinite(feat).all()


# ======================================================================================================
# input pre-processing: Pillow-BILINEAR resize + ToTensor + Normalize, bit-exact
# ======================================================================================================
@pytest.mark.parametrize("h,w,c,size,norm", [(97, 131, 3, 24, True), (19, 23, 3, 48, True), (301, 457, 3, 384, True), (640, 480, 3, 1024, True),
                                             (150, 111, 1, 32, False), (480, 640, 1, 384, False), (384, 384, 3, 384, True), (1024, 700, 3, 1024, True)])
def test_preprocess_bit_exact_vs_oracle(h, w, c, size, norm):
    """uint8 resize bit-identical to the oracle (itself pinned to Pillow); the float32 tensor bit-identical too (same IEEE ops:
    v / 255, then (x - mean) / std). ref: utils/dataloader.py:266-293."""
    from cor_amd import preprocess as CP
    from oracle import preprocess as OP
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    img = a if c == 3 else a[:, :, 0]
    want_u8 = OP.resize_bilinear_u8(img, size, size)
    want = OP.to_tensor_normalize(want_u8, OP.IMAGENET_MEAN if norm else None, OP.IMAGENET_STD if norm else None)
    t = torch.from_numpy(img).to(DEV)
    got, got_u8 = CP.resize_to_tensor(t, size, size, CP.IMAGENET_MEAN if norm else None, CP.IMAGENET_STD if norm else None, return_u8=True)
    gu = got_u8.cpu().numpy()
    assert np.array_equal(gu if c == 3 else gu[:, :, 0], want_u8)
    assert got.shape == (c, size, size) and np.array_equal(got.cpu().numpy(), want)


def test_preprocess_golden_pillow_and_transforms():
    """The committed Pillow outputs through the C ABI; the transform classes mirror the reference's Compose objects."""
    import os
    from cor_amd import preprocess as CP
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess_resize.npz"))
    for k in d.files:
        if not k.endswith("_in"):
            continue
        want = d[k[:-3] + "_out"]
        _, u8 = CP.resize_to_tensor(torch.from_numpy(d[k]).to(DEV), want.shape[0], want.shape[1], return_u8=True)
        u8 = u8.cpu().numpy()
        assert np.array_equal(u8 if want.ndim == 3 else u8[:, :, 0], want), k
    img = torch.randint(0, 256, (500, 375, 3), dtype=torch.uint8, device=DEV)
    assert CP.QueryImageTransform()(img).shape == (3, 1024, 1024)
    assert CP.ImageTransform(384)(img).shape == (3, 384, 384)
    assert CP.MaskTransform(384)(img[:, :, 0].contiguous()).shape == (1, 384, 384)
    with pytest.raises(RuntimeError):
        CP.ImageTransform(384)(img.cpu())


# ======================================================================================================
# BASELINE.json configs[1] (SAM-B + SigLIP-B/16-384, batch 32) and configs[3] (SAM-L + SigLIP-L/16-384, batch 64) at FULL size
# through size-independent properties
# ======================================================================================================
@pytest.mark.parametrize("sam_name,siglip_name,B,mode", [("sam_base", "ViT-B-16-SigLIP-384", 32, "bf16"), ("sam_base", "ViT-B-16-SigLIP-384", 8, "f32"),
                                                         ("sam_large", "ViT-L-16-SigLIP-384", 64, "bf16"),      # configs[1] (x2 modes), configs[3]
                                                         ("sam_huge", "ViT-SO400M-14-SigLIP-384", 16, "bf16")])  # largest SAM + the factory's DEFAULT tower (lib/build_model.py:14-20,43-47): head_dim 80 / 72 attention, K = 588 / 4304 GEMMs, 729 tokens
def test_full_size_batch_invariance_and_retrieval(sam_name, siglip_name, B, mode):
    """The oracle cannot run 32 full-size triplets in seconds, so the full configuration is checked through properties:
    (1) batch invariance: every sample of a batch-32 forward (persistent 256x256 GEMM kernel, pipelined attention) equals, bit
        for bit, the same sample run alone (batch 1: 128x128 GEMM kernels) - masks, image embeddings, fused features;
    (2) comb_support_feat rows are unit vectors (support_branch.py:85);
    (3) retrieval against a 10k-row gallery: sorted scores; sharding the gallery in two and merging on the host gives the
        identical top-k (scores and indices bitwise); the result is bitwise the CPU chain oracle's (fp32 and bf16 galleries)."""
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    from cor_amd import utils, retrieval
    T = torch.bfloat16 if mode == "bf16" else torch.float32
    model = build_model_with_query_support_feat(sam_name, siglip_name, None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=3)
    model = model.to(DEV).eval()
    model.compute_dtype = T
    batch = utils.synthetic_batch(B, DEV, seed=11)
    masks, emb, feat = model(**batch, multimask_output=True)
    assert masks.shape == (B, 1, 256, 256) and emb.shape == (B, 256, 64, 64) and feat.shape == (B, 1, 256)
    assert torch.isfinite(masks).all() and torch.isfinite(emb).all() and torch.isfinite(feat).all()
    for i in (0, B // 2 + 1, B - 1):
        one = {k: v[i:i + 1].contiguous() for k, v in batch.items()}
        m1, e1, f1 = model(**one, multimask_output=True)
        assert torch.equal(m1[0], masks[i]) and torch.equal(e1[0], emb[i]) and torch.equal(f1[0], feat[i]), f"sample {i} differs from its batch-1 run"
    q = feat[:, 0]
    assert torch.allclose(q.norm(dim=-1), torch.ones(B, device=DEV), atol=1e-5)
    gen = torch.Generator(device="cpu").manual_seed(5)
    rows = torch.nn.functional.normalize(torch.randn((10000, 256), generator=gen), dim=-1).to(DEV)
    where = torch.arange(B, device=DEV) * 151 + 7                            # < 10000 for B <= 64
    rows[where] = q                                  # (a random-init model's queries are nearly collinear: no self-match claim)
    gdt = torch.float32 if mode == "f32" else torch.bfloat16
    s_all, i_all = retrieval.GalleryShard(rows, 0, gdt).search(q, 10)
    assert bool((s_all[:, :-1] >= s_all[:, 1:]).all())                       # sorted
    # every gallery dtype: bitwise the CPU chain oracle (16-bit: over the stored values, query rounded to the gallery dtype)
    qr = q.cpu() if mode == "f32" else q.cpu().to(gdt).float()
    rs, ri = oret.similarity_topk_chain(qr, rows.cpu().to(gdt).float(), 10)
    assert torch.equal(i_all.cpu(), ri) and torch.equal(s_all.cpu().view(torch.int32), rs.view(torch.int32))
    sa, ia = retrieval.GalleryShard(rows[:5000], 0, gdt).search(q, 10)
    sb, ib = retrieval.GalleryShard(rows[5000:], 5000, gdt).search(q, 10)
    sm, im = retrieval.merge_topk_host([sa.cpu(), sb.cpu()], [ia.cpu(), ib.cpu()], 10)
    assert torch.equal(im, i_all.cpu()) and torch.equal(sm, s_all.cpu())


def test_reverse_work_order_is_bit_identical():
    """COR_ORDER_REVERSE (gemm cfg / layernorm act / sam_attention variant) only changes the order in which tiles, rows and
    windows are visited: outputs must be bit-identical (persistent and 128x128 GEMM kernels, bf16 and fp32, in-place residual)."""
    ops, _ = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    for (M, N, K, T, TO) in ((131072 // 8, 768, 768, BF16, F32), (2048 + 77, 2304, 768, BF16, BF16), (1000, 300, 256, F32, F32), (70000, 768, 3072, BF16, F32)):
        a = torch.randn((M, K), generator=g, device=DEV).to(T)
        w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(T)
        bias = torch.randn((N,), generator=g, device=DEV)
        res = torch.randn((M, N), generator=g, device=DEV) if TO == F32 else None
        o0 = ops.gemm(a, w, out_dtype=TO, bias=bias, residual=res)
        o1 = ops.gemm(a, w, out_dtype=TO, bias=bias, residual=res, reverse=True)
        assert torch.equal(o0, o1), (M, N, K)
        if res is not None:                                             # in place, as the encoder runs it
            x0, x1 = res.clone(), res.clone()
            ops.gemm(a, w, out_dtype=TO, bias=bias, residual=x0, out=x0)
            ops.gemm(a, w, out_dtype=TO, bias=bias, residual=x1, out=x1, reverse=True)
            assert torch.equal(x0, x1) and torch.equal(x0, o0)
    x = torch.randn((4099, 768), generator=g, device=DEV)
    wv, bv = torch.randn((768,), generator=g, device=DEV), torch.randn((768,), generator=g, device=DEV)
    assert torch.equal(ops.layernorm(x, wv, bv, 1e-6, out_dtype=BF16), ops.layernorm(x, wv, bv, 1e-6, out_dtype=BF16, reverse=True))
    B, H = 2, 12
    qkv = torch.randn((B * 4096, 3 * H * 64), generator=g, device=DEV).to(BF16)
    pad = torch.randn((3 * H * 64,), generator=g, device=DEV).to(BF16)
    for win, S in ((0, 64), (14, 14)):
        rh = torch.randn((2 * S - 1, 64), generator=g, device=DEV) * 0.3
        rw = torch.randn((2 * S - 1, 64), generator=g, device=DEV) * 0.3
        for variant in (0, 1):
            assert torch.equal(ops.sam_attention(qkv, pad, rh, rw, B, H, 64, win, variant=variant),
                               ops.sam_attention(qkv, pad, rh, rw, B, H, 64, win, variant=variant, reverse=True)), (win, variant)
        if win == 0:                                      # the pre-scaled forms of the global kernel (what the engine runs; the 64-query-per-wave form)
            from cor_amd import _native as nat
            c = nat.Q_PRESCALE_HD64
            for variant in (0, 4):
                assert torch.equal(ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, variant=variant, q_prescale=c),
                                   ops.sam_attention(qkv, pad, rh, rw, B, H, 64, 0, variant=variant, q_prescale=c, reverse=True)), variant


@pytest.mark.parametrize("mode", [F32, BF16])
def test_graph_captured_forward_equals_eager(mode):
    """model.capture(): the forward replayed as one hipGraph returns bit-identical outputs to the eager forward, for the
    captured inputs and for NEW inputs copied into the captured buffers; stale captures (parameters changed) are refused."""
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    from cor_amd import utils
    model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=3)
    model = model.to(DEV).eval()
    model.compute_dtype = mode
    b0 = utils.synthetic_batch(2, torch.device(DEV), seed=0)
    b1 = utils.synthetic_batch(2, torch.device(DEV), seed=1)
    e0 = [t.clone() for t in model(**b0, multimask_output=True)]
    e1 = [t.clone() for t in model(**b1, multimask_output=True)]
    g_serial = model.capture(**b0, multimask_output=True, overlap_branches=False)
    for a, b in zip(g_serial(**b1, clone=True), e1):
        assert torch.equal(a, b)
    del g_serial
    g = model.capture(**b0, multimask_output=True)                   # support branch as a parallel branch of the graph (default)
    for want, batch in ((e0, b0), (e1, b1), (e0, b0)):
        got = g(**batch, clone=True)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    # back to back WITHOUT host synchronisation between replays (bench.py awaits step i's results after enqueuing step i + 1): the
    # parallel support branch of replay i + 1 must not start on replay i's buffers - every replay's cloned outputs (the clone is
    # enqueued right behind its replay) equal the synchronised ones, for alternating inputs
    outs = [g(**batch, clone=True) for batch in (b0, b1) * 6]
    torch.cuda.synchronize()
    for n, got in enumerate(outs):
        for a, b in zip(got, (e0, e1)[n % 2]):
            assert torch.equal(a, b), n
    host = {k: v.cpu() for k, v in b1.items()}                        # CPU tensors are copied over
    for a, b in zip(g(**host), e1):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        g(**utils.synthetic_batch(3, torch.device(DEV), seed=0))
    # the graph reads the packed weights by address: a no-op model.to(dev) clears the model's cache - the graph holds the
    # pack alive, the replay is still right and re-installs it (the next eager forward uses the SAME tensors); a cache that was
    # REBUILT meanwhile (new tensors the graph does not read) is refused
    W0 = g.W
    model.to(DEV)
    assert model._packed.get(mode) is None
    for a, b in zip(g(**b1, clone=True), e1):
        assert torch.equal(a, b)
    assert model._packed.get(mode) is W0 and model.packed(mode) is W0
    model.invalidate_packed()
    for a, b in zip(model(**b1, multimask_output=True), e1):         # eager forward re-packs: new tensors
        assert torch.equal(a, b)
    assert model._packed.get(mode) is not W0
    with pytest.raises(RuntimeError, match="rebuilt"):
        g(**b0)
    model._packed[mode] = W0                                          # (parameters unchanged: hand the captured pack back)
    for a, b in zip(g(**b0, clone=True), e0):
        assert torch.equal(a, b)
    with torch.no_grad():
        next(model.parameters()).add_(1.0)
    with pytest.raises(RuntimeError):
        g(**b0)


def test_forward_support_captured_on_a_forked_stream():
    """engine.forward_support(two_chains=True) called under SOMEONE ELSE's graph capture, on a stream that is itself a fork of the
    capture stream, without a caller-provided flat fork: a fork of a fork crashed hipGraph's capture_end in round 4
    (profiles/r05_capture_nested_fork_record.txt); the guard keeps both towers on the current stream. Capture + replay, bit-identical
    to the eager two-chain result."""
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    from cor_amd import utils, engine
    model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=5)
    model = model.to(DEV).eval()
    model.compute_dtype = BF16
    b = utils.synthetic_batch(2, torch.device(DEV), seed=2)
    W = model.packed(BF16)
    gcfg, mp = model.support_branch.siglip.cfg, model.support_branch.mask_pooling_name
    args = (b["support_image_inputs"], b["change_text_inputs"], b["support_mask_inputs"])
    want = engine.forward_support(W, gcfg, mp, BF16, *args, two_chains=True).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):                                 # a forked stream: forward_support must not fork again from it
            got = engine.forward_support(W, gcfg, mp, BF16, *args, two_chains=True, text_stream=None)
        main.wait_stream(side)
    got.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("depth,stagger", [(2, False), (3, False), (2, True)])
def test_forward_pipeline_equals_single_forwards(depth, stagger):
    """model.capture_pipeline(): `depth` captured forwards on `depth` streams, consecutive submits overlapping on the GPU (bench.py
    --inflight 2). Twelve submits of three alternating batches, NO host synchronisation in between, every slot's outputs cloned by
    `then` on the slot's stream: each equals the eager forward of its batch bit for bit; the in-place input route (next_inputs)
    too; the event marks the completion of the submit's work."""
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    from cor_amd import utils
    model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=5)
    model = model.to(DEV).eval()
    model.compute_dtype = BF16
    names = ("query_image_inputs", "support_image_inputs", "change_text_inputs", "support_mask_inputs")
    batches = [utils.synthetic_batch(2, torch.device(DEV), seed=s) for s in (0, 1, 2)]
    want = [[t.clone() for t in model(**b, multimask_output=True)] for b in batches]
    pipe = model.capture_pipeline(**batches[0], multimask_output=True, depth=depth, stagger=stagger)   # stagger: two graphs per slot, encoders chained by an event
    assert pipe.stagger == stagger and pipe.slots[0][0].split == stagger
    assert len(pipe.slots) == depth and len({st.cuda_stream for _, st in pipe.slots}) == depth
    clone = lambda out: [t.clone() for t in out]                      # noqa: E731 (enqueued on the slot's stream, behind the replay)
    got = [pipe.submit([batches[n % 3][k] for k in names], then=clone) for n in range(12)]
    got[-1][2].synchronize()
    torch.cuda.synchronize()
    for n, (_, res, ev) in enumerate(got):
        assert ev.query()
        for a, b in zip(res, want[n % 3]):
            assert torch.equal(a, b), n
    for n in range(2 * depth):                                        # inputs written in place into the next slot's buffers
        for dst, k in zip(pipe.next_inputs(), names):
            dst.copy_(batches[(n + 1) % 3][k])
        got.append(pipe.submit(None, then=clone))
    torch.cuda.synchronize()
    for n, (_, res, _) in enumerate(got[12:]):
        for a, b in zip(res, want[(n + 1) % 3]):
            assert torch.equal(a, b), n
    with pytest.raises(ValueError):
        model.capture_pipeline(**batches[0], depth=0)


def test_distributed_search_on_a_one_rank_rccl_group():
    """The multi-rank retrieval path (query all-gather, shard search over all slots, packed-list gather, host merge) on the REAL backend:
    a one-rank `nccl` (= RCCL) process group on this box's GPU with always_collective=True. One rank moves nothing over xGMI, but
    all_gather_into_tensor / gather run through RCCL on the device tensors this code builds (layouts, dtypes, the device-event timing
    marks of bench.py's `rccl` record) - which no CPU gloo test can show. dst=0 and dst=None, ragged max_local."""
    import socket
    import torch.distributed as dist
    from cor_amd import retrieval
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dev = torch.device(DEV)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(11)
        Q = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((5, 256), dtype=np.float32)), dim=-1).to(dev)
        G = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((12500, 256), dtype=np.float32)), dim=-1).to(BF16).to(dev)
        shard = retrieval.GalleryShard(G, offset=100)
        s0, i0 = shard.search(Q, 10)
        timing = []
        s1, i1 = retrieval.distributed_search(Q, shard, 10, max_local=8, timing=timing, always_collective=True)          # dst = 0: gather
        s2, i2 = retrieval.distributed_search(Q, shard, 10, max_local=8, dst=None, always_collective=True)                # all-gather of the lists
        pend = retrieval.distributed_search(Q, shard, 10, max_local=8, always_collective=True, defer=True)                # pinned copy behind an event
        pend_local = retrieval.distributed_search(Q, shard, 10, defer=True)                                               # the one-rank shortcut, deferred
        s3, i3 = pend.result(); s4, i4 = pend_local.result()
        torch.cuda.synchronize()
        for s_, i_ in ((s1, i1), (s2, i2), (s3, i3), (s4, i4)):
            assert torch.equal(i_, i0.cpu()) and torch.equal(s_.view(torch.int32), s0.cpu().view(torch.int32))
        t = retrieval.resolve_timing(timing)
        assert t["calls"] == 1 and t["collective_ms"] > 0 and t["search_ms"] > 0 and dist.get_backend() == "nccl"
        _note(name="distributed_search_one_rank_rccl", **t)
        # the deferred collective path never makes the HOST wait for the stream (round 4: `block[cap, 0] = float(n)` copied a host scalar
        # behind the forward, so every rank idled its GPU through the enqueue of the collectives): behind ~100 ms of queued GPU work the
        # call returns at once, for both forms of the second collective
        import time
        a = torch.randn((8192, 8192), device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(16):
            a @ a
        t1 = time.perf_counter()
        p1 = retrieval.distributed_search(Q, shard, 10, max_local=8, always_collective=True, defer=True)
        p2 = retrieval.distributed_search(Q, shard, 10, max_local=8, dst=None, always_collective=True, defer=True)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        assert (t3 - t0) > 0.05 and (t2 - t1) < 0.4 * (t3 - t0), (t1 - t0, t2 - t1, t3 - t2)
        for p_ in (p1, p2):
            assert torch.equal(p_.result()[1], i0.cpu())
        _note(name="distributed_search_host_time_behind_queued_work", queued_gpu_ms=(t3 - t0) * 1e3, host_ms_two_calls=(t2 - t1) * 1e3)
    finally:
        dist.destroy_process_group()


def test_bench_contract_smoke(tmp_path):
    """bench.py end to end at a reduced batch (the driver runs the default line at round end): ONE JSON line with the contract's keys,
    `roofline` and `cpu_baseline` objects, recall over all queries with the planted positives found by both pipelines."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--gallery", "20000", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "recall"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and 0 < d["roofline"]["frac"] < 1
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    rec = d["recall"]
    assert rec["queries"] == 4 and rec["oracle_recall_at_1_planted"] == 1.0 and rec["recall_at_1"] == 1.0 and rec["max_pairwise_cos"] < 0.9
    # exact-fp32 HIP pipeline end to end vs the CPU oracle end to end: top-k INDICES identical (north_star)
    assert rec["fp32_mode_topk_entries"] == 30 and rec["fp32_mode_topk_index_mismatches_vs_cpu_oracle"] == 0, rec
    assert d["fp32_mode_topk_index_mismatches_vs_cpu_oracle"] == 0 and d["cpu_oracle_recall_at_1"] == 1.0
    assert "clock" in d and d["config"]["multimask_output"] is True and "awaited and merged after step i + 1" in d["config"]["results"]
    assert d["config"]["forwards_in_flight"] == 2 and "capture_pipeline" in d["config"]["launch"] and len(d["step_done_ms"]) == 2   # the default


def test_bench_multi_rank_code_path_on_a_one_rank_rccl_group(tmp_path):
    """bench.py --rehearse-rccl 1: the N > 1 branch of the bench (nccl process group with device_id, barrier, all-gather / gather inside the
    step, max-reduce of the time on a device tensor, `rccl` record from device events) executed on RCCL with one rank."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-rccl", "1", "--batch", "2", "--gallery", "12500", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rc = d["rccl"]
    assert d["n_gpus"] == 1 and rc["backend"] == "nccl" and rc["world_size"] == 1 and rc["calls"] == 2 and rc["collective_ms"] > 0 and rc["search_ms"] > 0


def test_bench_gpus2_gloo_rehearsal_launches_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` ALONE (no torchrun around it, no WORLD_SIZE): the launcher starts two ranks that share
    this box's one GPU, the gallery is sharded two ways, queries are all-gathered, lists gathered and merged; the line says n_gpus 2
    and carries the `rccl` record (backend gloo here: a rehearsal of the code path, not of RCCL)."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "2", "--gallery", "5000",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["value"] > 0 and d["scaling"] == "weak"
    rc = d["rccl"]
    assert rc["world_size"] == 2 and rc["backend"] == "gloo" and rc["calls"] == 2 and rc["collective_ms"] > 0 and rc["search_ms"] > 0

