"""Retrieval end of the path: region embeddings, gallery similarity + top-k, gallery sharding across ranks.

The reference has no gallery / top-k code; the definitions come from its only region-vs-query
similarity, the training loss: region embedding = utils/loss_func.py:35-56 (mask_pooling), query = comb_support_feat
(lib/support_branch.py:85), score = F.cosine_similarity (utils/loss_func.py:84) = dot product of unit vectors.
Order: score descending, then global gallery index ascending.

Multi-GPU (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests):
the gallery is row-sharded, rank r owning rows [r*ceil(G/R), ...). One exchange step on the data path: an
all-gather of the [B_local, 256] query embeddings (<= 64 KB per rank: latency-bound). Each rank then scores
ALL queries against its shard with the HIP kernel (cor_similarity_topk), and the packed per-shard (score, global index)
lists go to one rank in ONE gather and are merged once on the host by (score desc, index asc).
"""
from __future__ import annotations

import torch

from . import ops


def region_embedding(embeddings: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """ref: utils/loss_func.py:35-56. embeddings f32[B,C,H,W] (query_image_embeddings), mask f32[B,1,h,w] in [0,1]
    -> f32[B,1,C] unit-norm. HIP: cor_bilinear (clamped) + cor_masked_pool (NCHW, L2-normalised)."""
    B, C, H, W = embeddings.shape
    emb = embeddings.to(torch.float32).contiguous()
    m = mask.to(torch.float32).contiguous()
    if tuple(m.shape[-2:]) != (H, W):
        m = ops.bilinear(m, H, W)
    return ops.masked_pool(emb, m, B, H * W, C, feat_nchw=True, clamp01=True, l2norm=True).view(B, 1, C)


def merge_topk_host(scores_parts, idx_parts, k: int):
    """Host merge of per-shard lists by (score desc, index asc). parts: lists of CPU tensors [Bq, k_i]."""
    s = torch.cat(scores_parts, dim=1)
    i = torch.cat(idx_parts, dim=1)
    i_key = torch.where(i < 0, torch.full_like(i, torch.iinfo(torch.int64).max), i)   # missing entries last
    o1 = torch.sort(i_key, dim=1, stable=True).indices
    s, i = torch.gather(s, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(s, 1, o2), torch.gather(i, 1, o2)


class GalleryShard:
    """Rows [offset, offset + n) of a unit-norm gallery, resident in HBM as fp32 / bf16 / fp16."""

    def __init__(self, rows: torch.Tensor, offset: int = 0, dtype: torch.dtype | None = None):
        if not rows.is_cuda:
            raise RuntimeError("GalleryShard lives in GPU memory (no CPU path)")
        self.rows = rows.to(dtype or rows.dtype).contiguous()
        self.offset = int(offset)

    def __len__(self):
        return self.rows.shape[0]

    def search(self, queries: torch.Tensor, k: int):
        """queries f32[Bq,C] (unit-norm) -> (scores f32[Bq,k], global idx i64[Bq,k]) on the GPU."""
        q = queries.reshape(-1, queries.shape[-1]).to(self.rows.device, torch.float32).contiguous()
        if self.rows.shape[0] == 0:                  # an empty shard (more ranks than gallery rows): all-missing lists, like Ng < k
            return (torch.full((q.shape[0], k), float("-inf"), device=q.device),
                    torch.full((q.shape[0], k), -1, dtype=torch.int64, device=q.device))
        with torch.cuda.device(self.rows.device):
            return ops.similarity_topk(q, self.rows, k, g_offset=self.offset)


def shard_bounds(n_rows: int, world: int, rank: int):
    per = -(-n_rows // world)
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def _pack_lists(s: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
    """(scores f32[B,k], idx i64[B,k]) -> ONE int32 tensor [B,k,3] (score bits, index lo, index hi): 12 B per entry."""
    out = torch.empty(s.shape + (3,), dtype=torch.int32, device=s.device)
    out[..., 0] = s.contiguous().view(torch.int32)
    out[..., 1:] = i.contiguous().view(torch.int32).view(i.shape + (2,))
    return out


def _unpack_lists(p: torch.Tensor):
    return p[..., 0].contiguous().view(torch.float32), p[..., 1:].contiguous().view(torch.int64).squeeze(-1)


class _Marks:
    """Time marks of one distributed_search call (bench.py's `rccl` record): device events on the CURRENT stream under RCCL
    (torch's collectives run on their own stream, but the blocking forms make the current stream wait for them, so events on the
    current stream bracket them), host clocks under gloo (whose collectives work on host copies and synchronise anyway)."""

    def __init__(self, dev, host):
        self.dev, self.host, self.t = dev, host, []

    def mark(self):
        if self.host:
            import time
            self.t.append(time.perf_counter())
        else:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.dev))
            self.t.append(ev)

    def ms(self, a, b):
        return (self.t[b] - self.t[a]) * 1e3 if self.host else self.t[a].elapsed_time(self.t[b])


def resolve_timing(timing: list) -> dict:
    """Means over the calls recorded in `timing` (after a device synchronisation): collective_ms = query all-gather + list gather,
    search_ms = this rank's shard against all query slots (+ packing; + the host copies under gloo)."""
    n = max(len(timing), 1)
    ag = sum(m.ms(0, 1) for m in timing) / n
    se = sum(m.ms(1, 2) for m in timing) / n
    ga = sum(m.ms(2, 3) for m in timing) / n
    return dict(calls=len(timing), allgather_ms=ag, search_ms=se, gather_ms=ga, collective_ms=ag + ga)


class PendingSearch:
    """A search whose lists are still on their way to the host (distributed_search(..., defer=True)): the ONE device-to-host copy
    went to pinned memory behind an event, so the caller can enqueue the next step's forward first and call result() afterwards -
    the GPU never waits for the host merge. result() -> (scores, idx) CPU tensors on `dst` (every rank for dst=None), (None, None)
    elsewhere; idempotent."""

    def __init__(self, finish=None, event=None, value=None):
        self._finish, self._event, self._value = finish, event, value

    def result(self):
        if self._finish is not None:
            if self._event is not None:
                self._event.synchronize()
            self._value, self._finish, self._event = self._finish(), None, None
        return self._value


def _to_host_async(t: torch.Tensor):
    """Device tensor -> pinned host tensor, non-blocking, + the event that marks the copy's completion on the current stream."""
    if not t.is_cuda:
        return t, None
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    return h, ev


def distributed_search(local_queries: torch.Tensor, shard: GalleryShard, k: int, group=None, max_local: int | None = None,
                       dst: int | None = 0, timing: list | None = None, always_collective: bool = False, defer: bool = False):
    """All ranks call this with their own queries [B_local, C] and their gallery shard.

    Two collectives in all, as BASELINE.json's north_star describes it:
      1. ONE all-gather of the query embeddings over RCCL/xGMI (<= 64 KB per rank: latency-bound). The payload is a
         fixed-size block [max_local + 1, C]: rows 0..B_local-1 are the queries, the rest zeros, the last row carries B_local,
         so a ragged last batch (B_local differing between ranks) is legal as long as every rank passes the same `max_local`
         (a configuration constant, e.g. the loader's batch size; default: this rank's B_local, i.e. equal batches);
      2. each rank scores ALL world * max_local query SLOTS against its shard (cor_similarity_topk; the slots beyond a rank's
         count are zero rows whose results are dropped at the end - nothing is compacted, so the counts never have to reach
         the host in the middle of the step) and the packed per-shard lists ([slots,k,3] int32 = 12 B per entry) go to rank
         `dst` in ONE gather, where they are merged ONCE on the host by (score desc, global index asc). dst=None: all-gather
         instead, every rank merges.
    Host synchronisation (backend nccl = RCCL): none on the ranks that return (None, None); on `dst` exactly one device-to-host
    copy at the very end (lists + per-rank counts in one buffer) - the next step's forward can be enqueued before it is awaited.
    Under the gloo REHEARSAL backend the collectives work on host copies, so every rank synchronises twice (queries, lists).
    `max_local` must be given (and equal on all ranks) whenever per-rank query counts can differ: the all-gather is fixed-size.
    timing: a list that receives one _Marks per call (resolve_timing() turns them into milliseconds after a synchronisation).
    always_collective: take the collective path even in a group of ONE rank (tests: the only way to put this code on RCCL with a
    single GPU - a one-rank nccl group still moves the device tensors through all_gather_into_tensor / gather).
    defer: return a PendingSearch instead of the tensors: the device-to-host copy is enqueued (pinned memory + event) and the host merge
    happens in its result() - call it after enqueuing the next step's forward, so that the GPU does not idle through the host's turn.
    Returns (scores f32[B_total,k], idx i64[B_total,k]) CPU tensors, queries ordered by rank, on rank `dst` (every rank
    for dst=None); (None, None) on the other ranks."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always_collective):
        s, i = shard.search(local_queries, k)
        if defer:
            both, ev = _to_host_async(_pack_lists(s, i))         # one copy instead of two
            return PendingSearch(lambda: _unpack_lists(both), ev)
        return s.cpu(), i.cpu()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q = local_queries.reshape(-1, local_queries.shape[-1]).to(torch.float32).contiguous()
    b_local, C = q.shape
    cap = int(max_local) if max_local is not None else b_local
    if b_local > cap:
        raise ValueError(f"distributed_search: {b_local} local queries exceed max_local={cap}")
    dev = shard.rows.device
    host_coll = dist.get_backend(group) == "gloo"               # CPU rehearsal backend: collectives on host copies
    cdev = torch.device("cpu") if host_coll else dev
    block = torch.zeros((cap + 1, C), dtype=torch.float32, device=cdev)
    block[:b_local] = q.to(cdev)
    block[cap, :1].fill_(float(b_local))                         # (a fill kernel; `block[cap, 0] = x` copies a host scalar: the host would wait for the stream)
    allb = torch.empty((world * (cap + 1), C), dtype=torch.float32, device=cdev)
    marks = _Marks(dev, host_coll) if timing is not None else None
    if marks:
        marks.mark()
    dist.all_gather_into_tensor(allb, block, group=group)        # collective 1 (RCCL over xGMI)
    if marks:
        marks.mark()
    allb = allb.view(world, cap + 1, C)
    slots = allb[:, :cap].reshape(world * cap, C).to(dev)        # every slot is scored; counts stay where they are
    s, i = shard.search(slots, k)                                # local shard vs ALL query slots
    packed = _pack_lists(s, i).to(cdev)
    if marks:
        marks.mark()
    if dst is None:
        parts = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=torch.int32, device=cdev)
        dist.all_gather_into_tensor(parts, packed, group=group)  # collective 2 (all ranks merge)
        parts = parts.view((world,) + tuple(packed.shape))
        if marks:
            marks.mark(); timing.append(marks)
    else:
        glist = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
        dist.gather(packed, glist, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)   # collective 2
        if marks:
            marks.mark(); timing.append(marks)
        if rank != dst:
            return PendingSearch(value=(None, None)) if defer else (None, None)
        parts = torch.stack(glist, dim=0)
    # the ONE device-to-host copy: lists of all shards + the per-rank counts (as int32) in one buffer
    counts_i = allb[:, cap, 0].to(torch.int32)
    flat = torch.cat([parts.reshape(-1), counts_i.to(parts.device)])
    pshape = tuple(parts.shape)

    def finish(host):
        counts = host[-world:].tolist()
        if any(c < 0 or c > cap for c in counts):
            raise RuntimeError(f"distributed_search: inconsistent per-rank query counts {counts} for max_local={cap}")
        ps, pi = _unpack_lists(host[:-world].view(pshape))       # [world(shard), world*cap(slot), k]
        keep = torch.cat([torch.arange(r * cap, r * cap + counts[r]) for r in range(world)])
        return merge_topk_host(list(ps[:, keep]), list(pi[:, keep]), k)

    if defer:
        host, ev = _to_host_async(flat)
        return PendingSearch(lambda: finish(host), ev)
    return finish(flat.cpu())


@torch.no_grad()
def build_gallery(model, batches, dtype=torch.float16):
    """Offline gallery builder: SAM image encoder over gallery images + region mask pooling
    (utils/loss_func.py:35-56) -> unit-norm rows [G, 256] in `dtype`, in input order.
    `batches` yields dicts with "query_img" f32[B,3,1024,1024] and "query_mask" f32[B,1,h,w] (the reference's loader
    field names, utils/dataloader.py:244-369). Only the encoder half of the forward runs (engine.sam_encoder)."""
    from . import engine
    rows = []
    T = model._resolve_dtype()
    W = model.packed(T)
    cfg = model.image_encoder.cfg
    g = cfg["img"] // cfg["patch"]
    for b in batches:
        img = b["query_img"].to(model.device, torch.float32).contiguous()
        B = img.shape[0]
        tok = engine.sam_encoder(W, img, cfg, T)                                       # [B*g*g, 256] fp32 tokens
        m = b["query_mask"].to(model.device, torch.float32).contiguous()
        if tuple(m.shape[-2:]) != (g, g):
            m = ops.bilinear(m, g, g)
        # tokens are channels-last: pool them directly (feat_nchw=False), clamp + L2-normalise as mask_pooling does
        rows.append(ops.masked_pool(tok, m, B, g * g, cfg["out"], feat_nchw=False, clamp01=True, l2norm=True).to(dtype))
    return torch.cat(rows, dim=0)


def save_gallery(path, rows, world=1):
    """On-disk format: <path>.shardNN.pt (rows of shard NN as a tensor) + <path>.manifest.json (row ranges)."""
    import json
    n = rows.shape[0]
    shards = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        torch.save(rows[lo:hi].cpu().contiguous(), f"{path}.shard{r:02d}.pt")
        shards.append(dict(rank=r, lo=lo, hi=hi, file=f"{path}.shard{r:02d}.pt"))
    with open(f"{path}.manifest.json", "w") as f:
        json.dump(dict(rows=n, dim=int(rows.shape[1]), dtype=str(rows.dtype), world=world, shards=shards), f, indent=1)


def load_gallery_shard(path, rank, device):
    import json
    man = json.load(open(f"{path}.manifest.json"))
    sh = man["shards"][rank]
    return GalleryShard(torch.load(sh["file"]).to(device), offset=sh["lo"])
