#!/usr/bin/env python3
"""The STATED CPU baseline (SURVEY 8d "CPU baseline timing", BASELINE.json configs[0]), recorded once per round.

    python tools/cpu_baseline.py --out profiles/r04_cpu_baseline.json        (on the GPU box's host cores; no GPU is touched)

Protocol, as SURVEY 8d / BASELINE.md state it: the CPU oracle (oracle/: the fp32 PyTorch-CPU restatement of the reference's forward,
pinned to reference-generated goldens; `kind` "port") on synthetic triplets whose RAW images are 224x224 U[0,1] (uint8) and are
resized first by the oracle's Pillow-exact path to 1024x1024 / 384x384 and normalised (utils/dataloader.py:266-293), SAM-B +
SigLIP-B/16-384 + MaskAdapterPooling, multimask_output as the reference ships it (config/vaild_config/vaild_config.yaml:13: false),
then region-embedding similarity + top-10 against a 1 000-row fp32 gallery; B = 1 and B = 4; torch.set_num_threads(min(cores of
this process, 16): the box's share of the host); 1 warm-up + 3 timed iterations each. The reference's call path this stands for: my_test.py:75-81
(build) and utils/vailder.py:416-424 (the forward under no_grad). This is test infrastructure: nothing in cor_amd imports it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_cpu_baseline.json"))
    ap.add_argument("--sam", default="sam_base")
    ap.add_argument("--siglip", default="ViT-B-16-SigLIP-384")
    ap.add_argument("--gallery", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--timed", type=int, default=3)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--threads", type=int, default=0, help="0: the cores this process may run on, at most 16 (a GPU box gives one GPU's share of the "
                    "host: 16 cores, while the affinity mask shows every core of the machine - more threads than that share only oversubscribe)")
    args = ap.parse_args()
    from oracle import config as ocfg, model as omodel, preprocess as OP, retrieval as oret
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(args.threads or min(cores, 16))
    sd = ocfg.random_state(ocfg.model_spec(args.sam, args.siglip, "MaskAdapterPooling"), seed=0)
    gen = torch.Generator().manual_seed(1234)
    G = torch.nn.functional.normalize(torch.randn((args.gallery, 256), generator=gen), dim=-1)
    rng = np.random.default_rng(0)
    res = dict(protocol="SURVEY 8d / configs[0]: raw 224x224 U[0,1] uint8 images -> Pillow-exact resize to 1024 / 384 + normalise (oracle.preprocess) -> "
                        f"oracle fp32 forward ({args.sam}+{args.siglip}+MaskAdapterPooling, multimask_output=False as shipped) -> similarity/top-{args.topk} vs a "
                        f"{args.gallery}-row fp32 gallery; 1 warm-up + {args.timed} timed iterations per batch size",
               kind="port", cores=cores, os_cpu_count=os.cpu_count(), threads=torch.get_num_threads(), torch=torch.__version__)
    for B in args.batches:
        raw_q = rng.integers(0, 256, (B, 224, 224, 3), dtype=np.uint8)
        raw_s = rng.integers(0, 256, (B, 224, 224, 3), dtype=np.uint8)
        raw_m = np.zeros((B, 224, 224), np.uint8)
        raw_m[:, 60:170, 50:180] = 255
        text = torch.ones((B, 64), dtype=torch.int64)
        text[:, :8] = torch.randint(2, 32000, (B, 8), generator=gen)
        t0 = time.perf_counter()
        q = torch.from_numpy(np.stack([OP.preprocess_image(raw_q[b], 1024) for b in range(B)]))
        s = torch.from_numpy(np.stack([OP.preprocess_image(raw_s[b], 384) for b in range(B)]))
        m = torch.from_numpy(np.stack([OP.preprocess_image(raw_m[b], 384, normalize=False) for b in range(B)]))
        t_resize = time.perf_counter() - t0
        times = []
        for it in range(1 + args.timed):
            t0 = time.perf_counter()
            with torch.no_grad():
                _, _, feat = omodel.forward(sd, args.sam, args.siglip, "MaskAdapterPooling", q, s, text, m, False)
                oret.similarity_topk(feat[:, 0], G, args.topk)
            times.append(time.perf_counter() - t0)
            print(f"B={B} iteration {it}: {times[-1]:.2f} s", flush=True)
        timed = times[1:]
        mean = sum(timed) / len(timed)
        res[f"B{B}"] = dict(triplets_per_s=B / mean, s_per_step_mean=mean, s_per_step=timed, warmup_s=times[0],
                            resize_s_oracle_numpy=t_resize, note="resize is the oracle's numpy restatement of Pillow (a checker, not a tuned loader): timed apart")
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
