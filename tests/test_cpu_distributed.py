"""world_size-2 gloo test (CPU, runs under -m "not gpu") of the multi-rank retrieval path: query all-gather,
per-shard scoring, result gather and host merge (cor_amd.retrieval.distributed_search). The HIP scoring kernel
needs a GPU, so each rank's shard is a test double whose .search() is the CPU oracle; everything else is the
product's code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _OracleShard:
    def __init__(self, rows, offset):
        self.rows, self.offset = rows, offset

    def search(self, queries, k):
        from oracle import retrieval as oret
        if self.rows.shape[0] == 0:                                   # empty shard (more ranks than gallery rows): all-missing lists
            return torch.full((queries.shape[0], k), float("-inf")), torch.full((queries.shape[0], k), -1, dtype=torch.int64)
        s, i = oret.similarity_topk(queries, self.rows, min(k, self.rows.shape[0]))
        i = i + self.offset
        if s.shape[1] < k:       # pad like the kernel does (score -inf, index -1)
            pad = k - s.shape[1]
            s = torch.cat([s, torch.full((s.shape[0], pad), float("-inf"))], 1)
            i = torch.cat([i, torch.full((i.shape[0], pad), -1, dtype=torch.int64)], 1)
        return s, i


def _worker(rank, world, port, G, Q_all, k, split, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cor_amd import retrieval
        lo, hi = retrieval.shard_bounds(G.shape[0], world, rank)
        shard = _OracleShard(G[lo:hi], lo)
        qlo, qhi = split[rank], split[rank + 1]                      # ragged per-rank batches are legal with max_local
        cap = max(split[r + 1] - split[r] for r in range(world))
        # (1) default: ONE gather to rank 0, merged once there; the other ranks get (None, None)
        s, i = retrieval.distributed_search(Q_all[qlo:qhi], shard, k, max_local=cap)
        if rank == 0:
            out["s"], out["i"] = s, i
        else:
            assert s is None and i is None
        # (2) dst=None: all-gather of the packed lists, every rank merges and must hold the same answer
        s2, i2 = retrieval.distributed_search(Q_all[qlo:qhi], shard, k, max_local=cap, dst=None)
        ref = [(out["s"], out["i"])] if rank == 0 else [None]
        dist.broadcast_object_list(ref, src=0)
        assert torch.equal(ref[0][1], i2) and torch.equal(ref[0][0], s2)
        # (2b) deferred form: the same answer out of PendingSearch.result(), (None, None) off the destination rank, idempotent
        pend = retrieval.distributed_search(Q_all[qlo:qhi], shard, k, max_local=cap, defer=True)
        s3, i3 = pend.result()
        if rank == 0:
            assert torch.equal(i3, out["i"]) and torch.equal(s3, out["s"]) and pend.result()[1] is i3
        else:
            assert s3 is None and i3 is None
        # (3) more local queries than max_local must raise BEFORE any collective (same on every rank here)
        with pytest.raises(ValueError):
            retrieval.distributed_search(Q_all, shard, k, max_local=1)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_pack_unpack_lists_roundtrip():
    from cor_amd import retrieval
    s = torch.tensor([[1.5, float("-inf")], [-0.0, 3e-7]])
    i = torch.tensor([[5, -1], [2 ** 40 + 3, -2]], dtype=torch.int64)
    s2, i2 = retrieval._unpack_lists(retrieval._pack_lists(s, i))
    assert torch.equal(s.view(torch.int32), s2.view(torch.int32)) and torch.equal(i, i2)


@pytest.mark.parametrize("Ng,k,split", [(1000, 10, (0, 2, 3, 5, 6)), (3, 4, (0, 2, 4, 6, 6)), (9, 10, (0, 6, 6, 6, 6))])
def test_distributed_search_world4_gloo_with_empty_shards_and_ranks(Ng, k, split):
    """Four ranks: ragged query counts (a rank with NO queries), and galleries so small that the last shard(s) are EMPTY
    (shard_bounds gives rank 3 rows [3,3) of a 3-row gallery): empty shards contribute (-inf, -1) lists and the merge must
    still equal the single-process oracle."""
    from oracle import retrieval as oret
    gen = torch.Generator().manual_seed(1)
    G = torch.nn.functional.normalize(torch.randn((Ng, 256), generator=gen), dim=-1)
    Q = torch.nn.functional.normalize(torch.randn((6, 256), generator=gen), dim=-1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(4, _free_port(), G, Q, k, split, out), nprocs=4, join=True)
    rs, ri = oret.similarity_topk(Q, G, min(k, Ng))
    kk = min(k, Ng)
    assert torch.equal(out["i"][:, :kk], ri) and torch.allclose(out["s"][:, :kk], rs, atol=1e-6)   # (shard-sized vs full-size CPU matmuls)
    if k > Ng:
        assert (out["i"][:, Ng:] == -1).all()


@pytest.mark.parametrize("Ng,k,split", [(1000, 10, (0, 3, 6)), (7, 8, (0, 3, 6)), (1000, 10, (0, 4, 6)), (300, 5, (0, 6, 6))])
def test_distributed_search_world2_gloo(Ng, k, split):
    from oracle import retrieval as oret
    gen = torch.Generator().manual_seed(0)
    G = torch.nn.functional.normalize(torch.randn((Ng, 256), generator=gen), dim=-1)
    if Ng > 600:
        G[600] = G[3]                                   # a tie across the two shards
    Q = torch.nn.functional.normalize(torch.randn((6, 256), generator=gen), dim=-1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), G, Q, k, split, out), nprocs=2, join=True)
    rs, ri = oret.similarity_topk(Q, G, k)
    kk = min(k, Ng)
    assert torch.equal(out["i"][:, :kk], ri) and torch.allclose(out["s"][:, :kk], rs)
    if k > Ng:
        assert (out["i"][:, Ng:] == -1).all()
