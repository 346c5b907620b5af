#!/usr/bin/env python3
"""Where does the host block in distributed_search's collective path? One-rank nccl (= RCCL) group, a slow kernel in front of every call:
host time of each statement of the path (development probe; prints one JSON line)."""
import json, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    a = torch.randn((8192, 8192), device=dev)
    x = torch.randn((33, 256), device=dev)
    out = torch.empty((33, 256), device=dev)
    packed = torch.zeros((32, 10, 3), dtype=torch.int32, device=dev)
    res = {}

    def slow():
        for _ in range(20):
            a @ a                                   # ~tens of ms of GPU work in front of the statement under test

    def timed(name, fn):
        torch.cuda.synchronize()
        slow()
        t0 = time.perf_counter()
        r = fn()
        res[name] = round((time.perf_counter() - t0) * 1e3, 3)
        torch.cuda.synchronize()
        return r

    for rep in range(2):
        timed("slow_enqueue_only", lambda: None)
        timed("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out, x))
        timed("gather", lambda: dist.gather(packed, [torch.empty_like(packed)], dst=0))
        timed("setitem_float", lambda: x.__setitem__((32, 0), 5.0))
        timed("stack_cat", lambda: torch.cat([torch.stack([packed]).reshape(-1), x[:, 0].to(torch.int32)]))
        timed("pinned_empty", lambda: torch.empty((1000,), dtype=torch.int32, pin_memory=True))
        h = torch.empty((960,), dtype=torch.int32, pin_memory=True)
        timed("pinned_copy_async", lambda: h.copy_(packed.reshape(-1), non_blocking=True))
    t0 = time.perf_counter(); slow(); torch.cuda.synchronize(); res["slow_gpu_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
