#!/usr/bin/env python3
"""GPU micro-benchmark of cor_similarity_topk (gallery similarity + top-k) at the BASELINE shard shapes.
    python tools/sim_bench.py            # all shapes, one JSON line each (roofline of the similarity GEMM included)
    python tools/sim_bench.py 1m [reps]  # only 512 x 1M bf16, `reps` calls (for rocprofv3 --kernel-trace / --pmc passes)
Time = all launches of one call (small shards: block scan + wave selection; otherwise prep + sample scan + full scan + selection),
HIP events on the launch stream. Roofline: MFMA-bound when B_tot >= ~400 (2*Bq*256*Ng flop against 2.5 PF dense bf16), HBM-bound below
(Ng*256*2 B against 8 TB/s): the line reports both fractions."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cor_amd import ops
dev = "cuda:0"
SHAPES = ((32, 100000, 10, torch.bfloat16), (32, 12500, 10, torch.bfloat16), (256, 12500, 10, torch.bfloat16), (256, 100000, 10, torch.bfloat16), (512, 12500, 10, torch.bfloat16),
          (512, 32000, 10, torch.bfloat16),
          (512, 125000, 10, torch.bfloat16), (512, 125000, 10, torch.float16), (512, 125000, 10, torch.float32), (512, 1000000, 10, torch.float16),
          (512, 1000000, 10, torch.bfloat16))
only_1m = len(sys.argv) > 1 and sys.argv[1] == "1m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 4: timing-only ablation (results invalid); 8: never the two-launch small-shard path
if only_1m:
    SHAPES = ((512, 1000000, 10, torch.bfloat16),)
elif len(sys.argv) > 1 and "x" in sys.argv[1]:           # e.g. 512x12500 : one bf16 shape (for rocprofv3 passes)
    SHAPES = ((int(sys.argv[1].split("x")[0]), int(sys.argv[1].split("x")[1]), 10, torch.bfloat16),)
for Bq, Ng, k, dt in SHAPES:
    Q = torch.nn.functional.normalize(torch.randn((Bq, 256), device=dev), dim=-1)
    G = torch.nn.functional.normalize(torch.randn((Ng, 256), device=dev), dim=-1).to(dt)
    ts = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.similarity_topk(Q, G, k, flags=flags); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[1:])[len(ts[1:]) // 2] * 1e-3          # median of the warm calls, each timed alone (includes the host's launch latency
    # of the first kernel: the GPU is idle when the call starts). In the model step the stream is busy, so the figure that counts
    # is the device time per call when calls are enqueued back to back:
    NB = 8
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(NB):
        ops.similarity_topk(Q, G, k, flags=flags)
    e1.record(); e1.synchronize()
    tb = e0.elapsed_time(e1) / NB * 1e-3
    fl = 2.0 * Bq * Ng * 256
    gbytes = G.numel() * G.element_size()
    peak = 2500.0 if dt != torch.float32 else 157.3
    print(json.dumps(dict(Bq=Bq, Ng=Ng, k=k, dtype=str(dt), flags=flags, us_back_to_back=tb * 1e6, us_single_median=t * 1e6, us_single_min=min(ts[1:]) * 1e3,
                          tflops=fl / tb / 1e12,
                          roofline=dict(bound="mfma" if Bq >= 400 else "hbm", achieved=fl / tb / 1e12, peak=peak, unit="TFLOP/s", frac=fl / tb / 1e12 / peak,
                                        gallery_GBps=gbytes / tb / 1e9, hbm_frac=gbytes / tb / 8e12, algorithmic_bytes=gbytes + Bq * 1024,
                                        note="whole call (every launch of cor_similarity_topk), calls enqueued back to back"))), flush=True)
