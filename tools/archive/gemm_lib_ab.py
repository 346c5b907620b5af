#!/usr/bin/env python3
"""A/B of two BUILDS of the library on the batch-32 SAM-B block GEMMs (and optionally one bench step): alternating child processes,
one build each (a process can bind only one libcor). python tools/gemm_lib_ab.py <libA.so> <libB.so> [rounds]
Child mode: COR_AB_LIB=<path> python tools/gemm_lib_ab.py --child"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from cor_amd import _native
    _native.LIB_PATH = os.environ["COR_AB_LIB"]
    from cor_amd import ops
    M = 131072
    out = {}
    for N, K, mode, name in [(2304, 768, "plain", "qkv"), (768, 768, "res", "proj+res"), (3072, 768, "gelu", "lin1+gelu"), (768, 3072, "res", "lin2+res")]:
        A = torch.randn((M, K), device="cuda").to(torch.bfloat16); W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn((N,), device="cuda")
        x = torch.randn((M, N), device="cuda") if mode == "res" else None
        def run():
            if mode == "res":
                ops.gemm(A, W, out_dtype=torch.float32, bias=b, residual=x, out=x)
            else:
                ops.gemm(A, W, out_dtype=torch.bfloat16, bias=b, act=ops.ACT_GELU_ERF if mode == "gelu" else ops.ACT_NONE)
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out[name] = round(ts[len(ts) // 2], 1)
        del A, W, x
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child()
    else:
        libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
        rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
        for r in range(rounds):
            for lib in libs:
                p = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, COR_AB_LIB=lib), capture_output=True, text=True, timeout=300)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                print(json.dumps({"lib": os.path.relpath(lib, ROOT), "round": r, **(json.loads(line[-1]) if line else {"error": p.stderr[-300:]})}), flush=True)
