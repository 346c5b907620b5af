#!/usr/bin/env python3
"""GPU micro-benchmark of cor_gemm tile configurations (the per-call `cfg` argument) on the SAM-B / SigLIP-B shapes.
Interleaved rounds in ONE process, random data (cdna guide rules 24/25). Checks every configuration against
configuration 1 bit-for-bit tolerance-free on a sub-block (same accumulation order per k-step => tiny diffs only).
    python tools/gemm_bench.py [--cfgs 1 2 3 4 9 13] [--rounds 5]
The product library accepts the shipped selectors only (1, 2, 3, 4, 9, 13); anything else - the experimental kernels, selector 14, a
tile-order group (c >= 100) - needs the -DCOR_PROBES build (make -C cor_amd/csrc probes), which is bound automatically then.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops, _native

SHAPES = [  # (M, N, K, act, residual, out_f32, label)
    (131072, 2304, 768, 0, False, False, "sam qkv"),
    (131072, 768, 768, 0, True, True, "sam proj+res"),
    (131072, 3072, 768, 1, False, False, "sam lin1+gelu"),
    (131072, 768, 3072, 0, True, True, "sam lin2+res"),
    (18432, 2304, 768, 0, False, False, "siglip qkv"),
    (18432, 3072, 768, 1, False, False, "siglip fc1+gelu"),
    (131072, 256, 2304, 0, False, True, "neck 3x3"),
    (18432, 768, 768, 0, True, True, "siglip proj+res"),
    (18432, 768, 3072, 0, True, True, "siglip fc2+res"),
    (131072, 256, 768, 0, False, True, "neck 1x1"),
    (2048, 2304, 768, 0, False, False, "text qkv"),
    (8192, 1024, 1024, 0, True, True, "M=8192 square"),
    (2048, 3072, 768, 1, False, False, "text fc1+gelu"),
    (2048, 768, 3072, 0, True, True, "text fc2+res"),
    (2048, 768, 768, 0, True, True, "text proj+res"),
    (18432, 1024, 256, 1, False, False, "adapter 18432x1024x256"),
    (131072, 128, 256, 0, False, False, "decoder k/v proj N=128"),
    (131072, 256, 128, 0, True, True, "decoder out proj K=128"),
    (131072, 256, 256, 0, False, False, "decoder 256x256"),
]

def enc(c):
    """command-line cfg -> per-call cfg: kernel = c % 100, tile-order group = c // 100 (ablation bits 8.. of cfg)."""
    if isinstance(c, str) and ":" in c:                 # "13:0x4000": kernel 13 with development bits 0x4000 of the ablation field (probe build)
        k, d = c.split(":")
        return int(k) | (int(d, 0) << 8)
    c = int(c)
    return (c % 100) | ((16 * (c // 100)) << 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", nargs="*", default=["1", "2", "3", "4", "9", "13"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="", help="label substring filter")
    a = ap.parse_args()
    T = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    global SHAPES
    if a.only: SHAPES = [x for x in SHAPES if a.only in x[6]]
    if any(":" in c or int(c) >= 100 or (int(c) % 100) not in (0, 1, 2, 3, 4, 9, 13) for c in a.cfgs):
        _native.use_probe_library()                     # the production ABI answers COR_EINVAL to these
        print("# experimental selectors requested: bound tools/probes/libcor_probes.so", file=sys.stderr)
    lib = _native.load()
    dev = "cuda:0"
    res = []
    for (M, N, K, act, use_res, of32, label) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        A = torch.randn((M, K), generator=g, device=dev).to(T)
        W = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(T)
        bias = torch.randn((N,), generator=g, device=dev)
        R = torch.randn((M, N), generator=g, device=dev) if use_res else None
        od = torch.float32 if of32 else T
        outs, times = {}, {c: [] for c in a.cfgs}
        for c in a.cfgs:                                    # correctness + warm-up
            outs[c] = ops.gemm(A, W, out_dtype=od, bias=bias, act=act, residual=R, cfg=enc(c))[:512].float().clone()
        torch.cuda.synchronize()
        ref = outs[a.cfgs[0]]
        errs = {c: float((outs[c] - ref).abs().max()) for c in a.cfgs}
        tref = torch.nn.functional.linear(A[:512].float(), W.float(), bias)
        if act == 1: tref = torch.nn.functional.gelu(tref)
        if use_res: tref = tref + R[:512]
        err_ref = float((ref - tref).abs().max())
        for _ in range(a.rounds):
            for c in a.cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(A, W, out_dtype=od, bias=bias, act=act, residual=R, cfg=enc(c))
                e1.record(); e1.synchronize()
                times[c].append(e0.elapsed_time(e1))
        fl = 2.0 * M * N * K
        row = dict(shape=f"{M}x{N}x{K}", label=label, err_vs_torch=err_ref,
                   **{f"cfg{c}": dict(us_med=1e3 * sorted(times[c])[len(times[c]) // 2], tf_med=fl / (sorted(times[c])[len(times[c]) // 2] * 1e-3) / 1e12,
                                      tf_best=fl / (min(times[c]) * 1e-3) / 1e12, diff_vs_first=errs[c]) for c in a.cfgs})
        print(json.dumps(row), flush=True)
        res.append(row)

if __name__ == "__main__":
    main()
