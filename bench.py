#!/usr/bin/env python3
"""bench.py — CORE retrieval-time forward on MI355X: query triplets/sec (forward + gallery similarity + top-k).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...          # no WORLD_SIZE in the environment: this process starts the N ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Launching. `--gpus N` is honoured in every form: with WORLD_SIZE unset and N > 1 this process - which has
imported nothing that touches the GPU - starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free port> bench.py <same arguments>` as a CHILD, relays its output (rank 0's JSON line) and exits with its code;
with WORLD_SIZE set (the driver's torchrun form) it must equal --gpus, else the run refuses (exit 2) instead of printing N copies
of a 1-GPU number. The N > 1 line carries `rccl`: the backend and world size torch.distributed reports, and the mean
`collective_ms` (query all-gather + list gather) and `search_ms` of the steps, so that the scaling curve can be decomposed.

Workload (BASELINE.json metric: "query triplets/sec + Recall@1 vs 100k-region gallery"; model/batch of configs[1]):
SAM-ViT-B + SigLIP-B/16-384 + MaskAdapterPooling, batch 32 triplets per GPU, bf16 fast mode, synthetic inputs /
random-init weights, 100k-row bf16 gallery at every N (51 MB: fits one GPU). For N>1 (configs[2]) the gallery is
row-sharded over the ranks, queries are all-gathered over RCCL, each rank scores all queries against its shard, the packed
per-shard top-k lists go to rank 0 in one gather and are merged on the host. A "step" = one such pass; inputs resident in HBM.
The timed steps replay the forward as one captured hipGraph (`config.launch`; --graph 0 = eager launches); the GEMM events behind
`roofline` then come from an eager re-run of the same steps right after the timed region (`roofline.events`).
One JSON line on rank 0. `roofline` is for the dominant kernel (the bf16 MFMA GEMM); `cpu_baseline` is the CPU oracle
timed on the host cores on a bounded sample (3 triplets) at N=1; `recall` (outside the timed region) compares the benchmarked
pipeline's top-k over ALL 32 queries with the exact-fp32 HIP mode's (and the fp32 CPU oracle's on the 3 CPU triplets) against the
same gallery with one planted positive per query; inputs are per-sample texture tiles so that the 32 queries are not collinear.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch is imported in main(), AFTER the launch decision: the launching parent stays free of any GPU runtime.


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="triplets per GPU per step")
    ap.add_argument("--gallery", type=int, default=100000, help="total gallery rows (BASELINE metric: 100k-region gallery)")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--sam", default="sam_base")
    ap.add_argument("--siglip", default="ViT-B-16-SigLIP-384")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--gallery-dtype", default=None, choices=["bf16", "fp16", "f32"], help="storage type of the gallery rows (default: bf16 with --dtype bf16, f32 with --dtype f32)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4], help="BASELINE.json configs[N] as a preset (model, batch, gallery rows, gallery type; "
                    "explicit flags given after it still win): 1 = SigLIP-B/16 + SAM-B, batch 32, 10k gallery; 2 = the same with a 100k gallery sharded over --gpus ranks; "
                    "3 = SigLIP-L/16-384 + SAM-L, batch 64; 4 = the same with a 1M-row fp16 gallery sharded over --gpus ranks. 0 (default): configs[1]'s model and batch with "
                    "the metric's 100k-row gallery (the headline line)")
    ap.add_argument("--stagger", type=int, default=0, help="with --inflight > 1: 1 orders a slot's [encoder || support branch] graph behind the previous slot's by an "
                    "event (ForwardPipeline(stagger=True): measured equal at 4 hardware queues, slower at 8 / 16, profiles/r05_pipeline_queues.jsonl); 0 (default): "
                    "one graph per slot")
    ap.add_argument("--hw-queues", type=int, default=0, help="set GPU_MAX_HW_QUEUES before HIP initialises (0: leave the environment / runtime default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=-1, choices=[-1, 0, 1, 2], help="2: as 1, with the two SigLIP towers as two chains on two streams (round 4); 1: support branch (SigLIP towers, adapter, fusion) beside the SAM encoder: a parallel branch of "
                    "the captured graph / a second HIP stream in eager mode (+3-4 %% end to end). -1 (default): on under --graph 1, off in eager mode, "
                    "where concurrent kernels would inflate the per-launch GEMM events")
    ap.add_argument("--graph", type=int, default=1, help="1 (default): the timed steps replay the forward as ONE captured hipGraph "
                    "(model.capture: same kernels, lighter launch boundaries, +2.8 %% at batch 32, 1.4x at batch 1); ROCm cannot record "
                    "per-launch events inside a replayed graph, so the GEMM events behind `roofline` come from an eager re-run of the same "
                    "steps right after the timed region. 0: eager launches, events inside the timed region")
    ap.add_argument("--host-inputs", type=int, default=0, help="1: the batch starts in pinned host memory and is copied H2D inside every step "
                    "on a second stream, double-buffered (PCIe-inclusive rate for DESIGN.md; never the headline value)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl (= RCCL, default) | gloo (rehearsal: several ranks on ONE GPU)")
    ap.add_argument("--multimask", type=int, default=1, help="multimask_output of the forward (1: three masks + IoU arg-max select, as in rounds 1-3; "
                    "0: the reference's shipped config/vaild_config/vaild_config.yaml:13 - skips cor_iou_select's 3-way arg-max)")
    ap.add_argument("--defer", type=int, default=1, help="1 (default): a step's top-k lists go to pinned host memory behind an event and are awaited / merged on "
                    "the host AFTER the next step has been enqueued (the GPU does not idle through the host's turn; every step's result is on the host "
                    "before the timed region ends); 0: await each step's result before enqueuing the next (rounds 1-3)")
    ap.add_argument("--inflight", type=int, default=2, help="forwards in flight (with --graph 1 and resident inputs; 1: rounds 1-3): 2 (default) captures the forward twice (two sets of "
                    "buffers) and replays step i on HIP stream i %% 2, so the latency-bound tail of step i (support head, mask decoder, search: ~2.7 ms of small "
                    "kernels) runs beside the encoder GEMMs of step i + 1; every step's result still reaches the host inside the timed region")
    ap.add_argument("--pending", type=int, default=-1, help="results left un-awaited after an enqueue (-1: forwards in flight - 1, at least 1)")
    ap.add_argument("--rehearse-rccl", type=int, default=0, help="1 (with --gpus 1): run the N > 1 code path on a ONE-rank nccl (= RCCL) group - process-group "
                    "init with device_id, barrier, query all-gather, list gather, max-reduce of the time, the `rccl` record - the most of the multi-GPU "
                    "path one GPU can execute (tests); the line still says n_gpus 1")
    ap.add_argument("--launch-check", action="store_true", help="rehearse the launch path only: every rank joins the process group (gloo: no GPU "
                    "needed), all-reduces its rank and rank 0 prints one JSON line; nothing is benchmarked (tests/test_cpu_host.py)")
    args = ap.parse_args(argv)
    given = {a.split("=")[0] for a in (argv if argv is not None else sys.argv[1:]) if a.startswith("--")}
    if args.config:
        args.config_text = BASELINE_CONFIGS[args.config]
        for flag, attr, val in PRESETS[args.config]:
            if flag not in given:
                setattr(args, attr, val)
    else:
        args.config_text = None
    if args.gallery_dtype is None:
        args.gallery_dtype = "bf16" if args.dtype == "bf16" else "f32"
    return args


# BASELINE.json `configs`, verbatim, and what each means for this bench (the sharded ones shard over --gpus ranks)
BASELINE_CONFIGS = {
    1: "SigLIP-B/16 + SAM-ViT-B, batch 32, 10k gallery, 1\u00d7MI355X bf16",
    2: "SigLIP-B/16 + SAM-ViT-B, 100k gallery sharded 8 ways, RCCL all-gather over xGMI",
    3: "SigLIP-L/14 + SAM-ViT-L, 512\u00d7512 inputs, batch 64, 1\u00d7MI355X (HBM-bound attention tiles)",
    4: "SigLIP-L/14 + SAM-ViT-L, 1M-region gallery, fp16 embeddings, 8\u00d7MI355X with top-k merge",
}
_B = [("--sam", "sam", "sam_base"), ("--siglip", "siglip", "ViT-B-16-SigLIP-384"), ("--batch", "batch", 32)]
_L = [("--sam", "sam", "sam_large"), ("--siglip", "siglip", "ViT-L-16-SigLIP-384"), ("--batch", "batch", 64)]   # (the factory knows no L/14 tower: L/16-384 is its ViT-L)
PRESETS = {
    1: _B + [("--gallery", "gallery", 10000), ("--gallery-dtype", "gallery_dtype", "bf16")],
    2: _B + [("--gallery", "gallery", 100000), ("--gallery-dtype", "gallery_dtype", "bf16")],
    3: _L + [("--gallery", "gallery", 100000), ("--gallery-dtype", "gallery_dtype", "bf16")],
    4: _L + [("--gallery", "gallery", 1000000), ("--gallery-dtype", "gallery_dtype", "fp16")],
}


def launch_plan(gpus: int, env) -> tuple:
    """What this invocation must do, from --gpus and the environment alone (no GPU, no torch): ("run", None) - this process is a
    rank (or the single process of N = 1); ("spawn", N) - start N ranks as children; ("refuse", message) - WORLD_SIZE and --gpus
    disagree (a SCALE run must never print N copies of the 1-GPU number)."""
    if gpus < 1:
        return "refuse", f"--gpus {gpus}: need at least one GPU"
    ws = env.get("WORLD_SIZE")
    if ws is None:
        return ("run", None) if gpus == 1 else ("spawn", gpus)
    try:
        ws = int(ws)
    except ValueError:
        return "refuse", f"WORLD_SIZE={ws!r} is not an integer"
    if ws != gpus:
        return "refuse", f"WORLD_SIZE={ws} but --gpus {gpus}: launch one rank per GPU (or drop WORLD_SIZE and let bench.py start them)"
    return "run", None


def spawn_command(n: int, argv, port: int) -> list:
    """The child command of the ("spawn", N) plan: the driver's own torchrun form with this script's arguments unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n: int, argv) -> int:
    """Start the N ranks as ONE child (torchrun), relay its stdout line by line (rank 0's JSON line is the only line the ranks
    print there), return its exit code. The parent never initialises a GPU runtime, so no exec / fork hazard arises."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0))) // n)))
    proc = subprocess.Popen(spawn_command(n, argv, port), stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    got_line = False
    for line in proc.stdout:
        got_line = got_line or line.startswith("{")
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("bench.py: the ranks exited 0 without printing a result line", file=sys.stderr)
        return 3
    return rc


def launch_check(args):
    """--launch-check: the rank side of the launch path without the benchmark (CPU-only under gloo)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    total = rank
    if world > 1:
        if args.backend == "gloo":
            dist.init_process_group("gloo")
            t = torch.tensor([rank], dtype=torch.int64)
        else:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            t = torch.tensor([rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        total, got_world, backend = int(t.item()), dist.get_world_size(), dist.get_backend()
        dist.destroy_process_group()
    else:
        got_world, backend = 1, "none"
    if rank == 0:
        print(json.dumps({"launch_check": True, "gpus": args.gpus, "world_size": got_world, "backend": backend, "rank_sum": total}), flush=True)


NB_CPU = 3          # bounded CPU sample: 3 triplets ~ 14 s on 16 cores (the contract asks for 10-30 s)
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")   # written by tools/pmc_traffic.py for THIS build


def gemm_source_id():
    """sha256 over the GEMM kernel sources: a PMC traffic file is only quoted for the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "cor_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline_and_recall(args, model, batch, gallery_rows_cpu, dev, feat_fast):
    """(1) cpu_baseline: the CPU oracle (fp32 PyTorch-CPU restatement of the reference, pinned by tests/golden; kind "port")
    on the first NB_CPU triplets of the SAME synthetic batch with the SAME weights + similarity vs the same gallery.
    (2) recall, over ALL queries of the batch, outside every timed region. Reference features = the exact-fp32 HIP
    mode (pinned to the reference's goldens at 1e-5; the CPU oracle cannot run 32 full-size triplets in seconds) - and, for
    the first NB_CPU queries, the fp32 CPU oracle itself. Positives are planted from the reference features
    (gallery[pi(b)] = normalize(q_b + 0.1 N(0,1))); the benchmarked pipeline (feat_fast: the bf16 forward's features, then
    cor_similarity_topk on the bf16 gallery) must return the reference's top-1 (fp32 features, fp32 CPU product on the same rows)."""
    import torch
    from oracle import model as omodel, retrieval as oret
    from cor_amd import retrieval
    mm = bool(args.multimask)
    ncores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box gives one GPU's share of the host: 16 cores
    torch.set_num_threads(ncores)
    sd = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
    inp = {k: v[:NB_CPU].cpu() for k, v in batch.items()}
    gdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[args.gallery_dtype]
    G32 = gallery_rows_cpu.to(gdt).float()              # the stored rows, as the oracle sees them
    t0 = time.perf_counter()
    with torch.no_grad():
        _, _, feat = omodel.forward(sd, args.sam, args.siglip, "MaskAdapterPooling", inp["query_image_inputs"], inp["support_image_inputs"],
                                    inp["change_text_inputs"], inp["support_mask_inputs"], mm)
        oret.similarity_topk(feat[:, 0], G32, args.topk)
    dt = time.perf_counter() - t0
    cpu = dict(value=NB_CPU / dt, unit="triplets/s", cores=torch.get_num_threads(), kind="port",
               sample=f"{NB_CPU} triplets in one batch ({args.sam}+{args.siglip} fp32 forward + {G32.shape[0]}-row similarity/top-k), 1 run, {dt:.1f} s; "
                      "SURVEY 8d's protocol (B = 1 and 4, 1 warm-up + 3 timed) would take minutes of CPU time: bounded to one un-warmed batch")
    q_cpu = feat[:, 0].float()
    # ---- reference features of the whole batch: exact-fp32 HIP mode
    B = feat_fast.shape[0]
    fast_dtype = model.compute_dtype
    model.compute_dtype = torch.float32
    with torch.no_grad():
        q_ref = torch.cat([model(**{k: v[i:i + 8] for k, v in batch.items()}, multimask_output=mm)[2][:, 0] for i in range(0, B, 8)]).float().cpu()
    model.compute_dtype = fast_dtype
    cosm = q_ref.double() @ q_ref.double().T - 2 * torch.eye(B, dtype=torch.float64)
    gen = torch.Generator(device="cpu").manual_seed(4321)
    Gn = gallery_rows_cpu.shape[0]
    if Gn < 14 + B:
        raise ValueError(f"recall leg: a {Gn}-row gallery cannot hold {B} planted positives")
    where = 13 + torch.arange(B) * ((Gn - 14) // max(B, 1))                          # distinct, < Gn for every batch / gallery size
    assert int(where.max()) < Gn and where.unique().numel() == B
    G = gallery_rows_cpu.clone()
    G[where] = torch.nn.functional.normalize(q_ref + 0.1 * torch.randn(q_ref.shape, generator=gen), dim=-1)
    G = G.to(gdt)
    rs, ri = oret.similarity_topk(q_ref, G.float(), args.topk)                       # fp32 reference features, fp32 CPU product
    _, ri_cpu = oret.similarity_topk(q_cpu, G.float(), args.topk)                    # fp32 CPU oracle end to end (first NB_CPU queries)
    gshard = retrieval.GalleryShard(G.to(dev), 0)
    gs, gi = gshard.search(feat_fast.float().contiguous(), args.topk)
    gi = gi.cpu()
    # north_star's "bit-exact top-k indices" claim, end to end in the mode that can deliver it: exact-fp32 HIP forward +
    # cor_similarity_topk on the device  vs  fp32 CPU oracle forward + CPU chain top-k, same weights / inputs / gallery
    # (like with like: an fp32 gallery holding the stored bf16 values -> the device runs the exact fp32 fmaf chain, and so does the oracle;
    # a 16-bit gallery would round the QUERY to bf16 on the device, which the fp32 oracle does not)
    _, gi32 = retrieval.GalleryShard(G.float().to(dev), 0).search(q_ref[:NB_CPU].to(dev).contiguous(), args.topk)
    gi32 = gi32.cpu()
    _, ri_cpu = oret.similarity_topk(q_cpu, G.float(), args.topk, exact_chain=True)
    rec = dict(recall_at_1=float((gi[:, 0] == ri[:, 0]).float().mean()), queries=B,
               recall_at_1_planted=float((gi[:, 0] == where).float().mean()), oracle_recall_at_1_planted=float((ri[:, 0] == where).float().mean()),
               topk_index_mismatches=int((gi != ri).sum()), topk_entries=int(ri.numel()),
               max_pairwise_cos=float(cosm.max()), feature_max_abs_err=float((feat_fast.float().cpu() - q_ref).abs().max()),
               cpu_oracle_queries=NB_CPU, cpu_oracle_recall_at_1=float((gi[:NB_CPU, 0] == ri_cpu[:, 0]).float().mean()),
               cpu_oracle_recall_at_1_planted=float((ri_cpu[:, 0] == where[:NB_CPU]).float().mean()),
               cpu_oracle_vs_fp32_hip_feature_max_abs_err=float((q_cpu - q_ref[:NB_CPU]).abs().max()),
               fp32_mode_topk_index_mismatches_vs_cpu_oracle=int((gi32 != ri_cpu).sum()), fp32_mode_topk_entries=int(ri_cpu.numel()),
               fp32_mode_features_cpu_topk_index_mismatches_vs_cpu_oracle=int((ri[:NB_CPU] != ri_cpu).sum()),
               definition="top-1 of the benchmarked pipeline (bf16 forward + cor_similarity_topk, bf16 gallery) == top-1 of the reference (exact-fp32 HIP "
                          "features, pinned to the reference's goldens; fp32 CPU product) over all queries, gallery with one planted positive per query; "
                          "cpu_oracle_*: the same against the fp32 CPU oracle's own features for the first queries; fp32_mode_topk_index_mismatches_vs_cpu_oracle: "
                          "top-k INDICES of the exact-fp32 HIP pipeline end to end (fp32 forward + cor_similarity_topk) vs the fp32 CPU oracle end to end (its own "
                          "forward + chain top-k) on the first queries - north_star's bit-exact top-k claim in the mode that can deliver it (the exact-fp32 HIP mode "
                          "is pinned to the reference's goldens by tests/test_gpu_parity.py::test_batch32_bf16_vs_fp32_exact_mode_anchored_to_the_golden); inputs: per-sample texture tiles, "
                          "support-head biases zeroed (utils.synthetic_batch(structured=True), utils.zero_support_head_biases: de-collinearised queries)")
    return cpu, rec


def stated_cpu_baseline():
    """SURVEY 8d's stated CPU protocol (configs[0]: raw 224x224 images resized first, B = 1 and 4, 1k gallery, 1 warm-up + 3 timed),
    recorded once by tools/cpu_baseline.py on a GPU box's host cores; quoted beside the bounded sample when the file exists."""
    f = os.path.join(ROOT, "profiles", "r04_cpu_baseline.json")
    try:
        j = json.load(open(f))
        return dict(file=os.path.relpath(f, ROOT), **{k: j[k] for k in ("B1", "B4", "cores", "threads", "protocol") if k in j})
    except Exception:                                    # noqa: BLE001
        return None


def main():
    args = parse()
    plan, detail = launch_plan(args.gpus, os.environ)
    if plan == "refuse":
        print(f"bench.py: {detail}", file=sys.stderr)
        raise SystemExit(2)
    if plan == "spawn":
        raise SystemExit(spawn_ranks(detail, sys.argv[1:]))
    if args.launch_check:
        return launch_check(args)
    # HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default). With two forwards in flight the slots' streams share
    # queues, which staggers the forwards; with 8 or 16 queues a step is 1.6-2.2 % slower (profiles/r05_pipeline_queues.jsonl; the explicit
    # event-ordered form, --stagger 1, does not remove that). The measured configuration is the runtime's default, pinned here; --hw-queues N
    # overrides it for A/B runs; capture_pipeline warns when the environment disagrees.
    os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues) if args.hw_queues > 0 else os.environ.get("GPU_MAX_HW_QUEUES", "4")
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = 0                                   # rehearsal: all ranks share the one visible GPU
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} over RCCL needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         "(--backend gloo rehearses several ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    from cor_amd import engine, ops, retrieval, utils
    overlap = {0: False, 1: "one_chain", 2: True}[(2 if args.graph else 0) if args.overlap < 0 else args.overlap]   # True: the two SigLIP towers as two chains
    engine.OVERLAP_BRANCHES = overlap and not args.graph   # eager steps; the captured graph takes it as an argument
    from cor_amd.lib.build_model import build_model_with_query_support_feat

    T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    mm = bool(args.multimask)
    model = build_model_with_query_support_feat(args.sam, args.siglip, None, None, "MaskAdapterPooling")
    utils.randomize_parameters(model, seed=0)
    utils.zero_support_head_biases(model)               # with structured inputs: 32 distinct query embeddings instead of 32 collinear ones
    model = model.to(dev).eval()
    model.compute_dtype = T
    B = args.batch
    batch = utils.synthetic_batch(B, dev, seed=rank, structured=True)
    Gtot = args.gallery
    lo, hi = retrieval.shard_bounds(Gtot, world, rank)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    rows_all = torch.nn.functional.normalize(torch.randn((Gtot, 256), generator=gen), dim=-1)
    gdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[args.gallery_dtype]
    shard = retrieval.GalleryShard(rows_all[lo:hi].to(dev), offset=lo, dtype=gdt)

    host_batch = {k: v.cpu().pin_memory() for k, v in batch.items()} if args.host_inputs else None
    h2d = utils.DoubleBufferedH2D(batch, dev) if args.host_inputs else None     # copy of batch i+1 under the forward of batch i
    if h2d is not None:
        h2d.stage(host_batch)

    graphed, launch_mode = None, "eager ctypes launches"     # (captured BEFORE the process group exists: no RCCL host threads beside the capture)
    names = ("query_image_inputs", "support_image_inputs", "change_text_inputs", "support_mask_inputs")
    pipe = None                                          # --inflight > 1: cor_amd's ForwardPipeline (N captured forwards on N streams)
    if args.graph:
        try:
            if args.inflight > 1 and not args.host_inputs:
                pipe = model.capture_pipeline(**batch, multimask_output=mm, depth=args.inflight, overlap_branches=overlap, stagger=bool(args.stagger))
                graphed = pipe.slots[0][0]
            else:
                graphed = model.capture(**batch, multimask_output=mm, overlap_branches=overlap)
            if not args.host_inputs:                     # resident inputs: the batch IS the graph's input buffers (no per-step D2D copy)
                batch = dict(zip(names, graphed.static_in))
            launch_mode = "hipGraph replay of the forward (model.capture" + (", support branch as a parallel graph branch" if overlap else "") + "); similarity search eager"
            if pipe is not None:
                launch_mode += (f"; {args.inflight} forwards in flight (model.capture_pipeline: step i replays captured graph i % {args.inflight}, which has its own "
                                f"buffers, on stream i % {args.inflight}, followed there by its search and the lists' copy to the host"
                                + ("; the slots' [encoder || support branch] graphs are chained by an event, the decoder graph of step i runs beside step i + 1's encoder)" if pipe.stagger else ")"))
        except Exception as e:                           # noqa: BLE001 - the bench must still produce its line
            graphed, pipe = None, None
            launch_mode = f"eager ctypes launches (graph capture failed: {type(e).__name__}: {e})"

    rehearse = bool(args.rehearse_rccl) and world == 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL on ROCm
    elif rehearse:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    multi = world > 1 or rehearse                        # the collective code path runs

    def search(out):
        # defer: the lists' device-to-host copy is enqueued behind an event; the host merge of step i runs after step i + 1 is enqueued
        return retrieval.distributed_search(out[2][:, 0], shard, args.topk, max_local=B, timing=timing, always_collective=rehearse, defer=True)   # (awaited at once under --defer 0)

    def step():
        if pipe is not None:                             # every slot's input buffers hold the synthetic batch (resident inputs)
            return pipe.submit(None, then=search)[1]
        if h2d is not None:
            b = h2d.take()
            h2d.stage(host_batch)                      # next step's inputs start crossing PCIe now, on the copy stream
        else:
            b = batch
        return search(graphed(**b) if graphed is not None else model(**b, multimask_output=mm))

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    timing = None
    for _ in range(args.warmup):
        step().result()
    barrier()
    prof = []
    timing = [] if multi else None                       # per-step marks around the two collectives and the shard search
    clock = utils.ClockSampler(dev).start()
    if graphed is None:
        ops.GEMM_PROFILE = prof                          # HIP events around every cor_gemm, on the launch stream
    t0 = time.perf_counter()
    pend = collections.deque()
    keep = (args.pending if args.pending >= 0 else max(1, (args.inflight if pipe is not None else 1) - 1)) if args.defer else 0   # results not yet awaited after an enqueue (1: step i - 1 is awaited once step i is enqueued)
    done_at = []                                         # host time at which each step's merged top-k was in hand
    for _ in range(args.steps):
        pend.append(step())                              # forward + search of this step are enqueued ...
        while len(pend) > keep:
            out = pend.popleft().result()                # ... before an earlier step's lists are awaited and merged on the host
            done_at.append(time.perf_counter() - t0)
    while pend:
        out = pend.popleft().result()                    # every step's top-k is materialised on the host inside the timed region
        done_at.append(time.perf_counter() - t0)
    barrier()
    dt = time.perf_counter() - t0
    clk = clock.stop()
    ops.GEMM_PROFILE = None
    coll = retrieval.resolve_timing(timing) if timing else None
    timing = None
    events_from = "the timed region"
    if graphed is not None:
        # the same kernels, launched eagerly so that events can bracket every GEMM (not part of `value`)
        events_from = "an eager re-run of the same steps right after the timed region (no per-launch events inside a replayed hipGraph on ROCm)"
        ops.GEMM_PROFILE = prof
        engine.OVERLAP_BRANCHES = False                  # one stream: a GEMM's events then bracket that GEMM alone
        for _ in range(args.steps):
            model(**batch, multimask_output=mm)
        torch.cuda.synchronize()
        ops.GEMM_PROFILE = None
    if multi:
        tmax = torch.tensor([dt], device="cpu" if (args.backend == "gloo" and world > 1) else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # dominant kernel: the MFMA GEMM. HIP events were recorded around every launch on the launch stream.
    gemm_ms = sum(p[0].elapsed_time(p[1]) for p in prof if p[3] == T)
    gemm_flops = sum(p[2] for p in prof if p[3] == T)
    gemm_bytes = sum(p[4] for p in prof if p[3] == T)
    n_launch = sum(1 for p in prof if p[3] == T)
    # HBM traffic of that kernel cannot be read live: it comes from the committed rocprofv3 --pmc passes of THIS command
    # (tools/pmc_traffic.py -> profiles/*pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction), per launch.
    traffic, traffic_note, traffic_ratio = None, "no PMC pass for this build", None
    lps = n_launch // max(args.steps, 1)
    if os.path.exists(PMC_TRAFFIC_FILE) and args.dtype == "bf16" and world == 1 and B == 32 and args.sam == "sam_base":
        try:
            tj = json.load(open(PMC_TRAFFIC_FILE))
            rel = os.path.relpath(PMC_TRAFFIC_FILE, ROOT)
            if tj.get("gemm_source_id") != gemm_source_id():
                traffic_note = f"{rel} was measured on other GEMM sources ({tj.get('gemm_source_id')}): not quoted"
            elif tj.get("launches_per_step") != lps:
                # same denominators or nothing (a 215-launch PMC pass was quoted beside a 200-launch step)
                traffic_note = f"{rel} counts {tj.get('launches_per_step')} GEMM launches per step, this run has {lps}: not quoted"
            else:
                traffic = tj["bytes_per_step"] / lps
                traffic_ratio = tj["bytes_per_step"] / (gemm_bytes / max(args.steps, 1))
                traffic_note = f"{rel} (same GEMM sources {tj['gemm_source_id']}, same {lps} launches per step; bytes per step {tj['bytes_per_step']:.4g})"
        except Exception as e:                           # noqa: BLE001
            traffic_note = f"unreadable PMC file: {e}"
    peak = 2500.0 if T == torch.bfloat16 else 157.3
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0

    if rank == 0:
        res = {
            "metric": "query triplets/sec (forward + similarity + top-k)", "value": world * B * args.steps / dt, "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if not args.host_inputs else "synthetic, inputs copied from pinned host memory inside the step",
            "config": {"workload": (f"BASELINE configs[{args.config}]: {args.config_text} -> " if args.config else "")
                                   + f"{args.sam}+{args.siglip}+MaskAdapterPooling, {B} triplets/GPU, {Gtot}-row {args.gallery_dtype} gallery"
                                   + (f" sharded {world} ways ({'RCCL' if args.backend == 'nccl' else 'gloo REHEARSAL on one GPU:'} all-gather of queries, host top-k merge)" if world > 1 else ""),
                       "global_batch": world * B, "gallery_rows": Gtot, "topk": args.topk, "parallelism": f"dp{world}+gallery-shard{world}"},
            "roofline": {"bound": "mfma", "kernel": ("cor_gemm, bf16 operands: gemm_pp<*> (persistent 256x256 ping-pong, >= 140 tiles) + gemm_tile<bf16,*,128,128> (the rest)" if args.dtype == "bf16" else "cor_gemm, fp32 operands: gemm_tile<float,float,128,128> on v_mfma_f32_32x32x2_f32"), "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes per launch (PMC, offline pass)", "traffic_source": traffic_note,
                         "traffic_over_algorithmic": traffic_ratio,
                         "algorithmic_bytes_per_launch": gemm_bytes / max(n_launch, 1), "launches_per_step": n_launch // max(args.steps, 1),
                         "avg_launch_us": gemm_ms * 1e3 / max(n_launch, 1), "gemm_share_of_step": gemm_ms / (dt * 1e3)},
        }
        res["config"]["launch"] = launch_mode
        res["config"]["multimask_output"] = mm
        res["config"]["forwards_in_flight"] = args.inflight if pipe is not None else 1
        res["config"]["hip_hw_queues"] = int(os.environ["GPU_MAX_HW_QUEUES"])
        res["config"]["pipeline_stagger"] = bool(pipe.stagger) if pipe is not None else None
        res["config"]["gallery_dtype"] = args.gallery_dtype
        res["config"]["results"] = ("every step's top-k lists reach the host inside the timed region; step i's lists are awaited and merged after step i + 1 "
                                    "is enqueued (pinned copy behind an event)") if args.defer else "each step's top-k lists are awaited before the next step is enqueued"
        res["roofline"]["events"] = events_from
        res["clock"] = clk
        res["step_done_ms"] = [round(t * 1e3, 2) for t in done_at]   # with two forwards in flight the first result arrives late, then one per steady-state interval
        if multi:
            res["rccl"] = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), **coll,
                               note="rank 0's means over the timed steps; device events on the launch stream under nccl (= RCCL), host clocks under the "
                                    "gloo rehearsal (whose collectives work on host copies); collective_ms = query all-gather + packed-list gather")
        if world == 1 and not args.no_cpu_baseline:
            try:                                         # a failure in the checker legs must not discard the measured throughput
                with torch.no_grad():                    # the benchmarked pipeline's features of this batch (graph replay when --graph 1)
                    feat_fast = (graphed(**batch) if graphed is not None else model(**batch, multimask_output=mm))[2][:, 0].clone()
                res["cpu_baseline"], rec = cpu_baseline_and_recall(args, model, batch, rows_all, dev, feat_fast)
                stated = stated_cpu_baseline()
                if stated:
                    res["cpu_baseline"]["stated_protocol"] = stated
                res["recall_at_1"] = rec["recall_at_1"]
                res["cpu_oracle_recall_at_1"] = rec["cpu_oracle_recall_at_1"]
                res["fp32_mode_topk_index_mismatches_vs_cpu_oracle"] = rec["fp32_mode_topk_index_mismatches_vs_cpu_oracle"]
                res["recall"] = rec
            except Exception as e:                       # noqa: BLE001
                res["cpu_baseline"] = None
                res["recall"] = dict(error=f"{type(e).__name__}: {e}")
        print(json.dumps(res), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
