import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/cor_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
from cor_amd import ops, _native
_native.use_probe_library()
dev = "cuda:0"
Bq, Ng = 512, 1000000
for dt in (torch.bfloat16, torch.float16):
    Q = torch.nn.functional.normalize(torch.randn((Bq, 256), device=dev), dim=-1)
    G = torch.nn.functional.normalize(torch.randn((Ng, 256), device=dev), dim=-1).to(dt)
    for rnd in range(3):
        for fl in (4, 4 | 512):
            for _ in range(2): ops.similarity_topk(Q, G, 10, flags=fl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ops.similarity_topk(Q, G, 10, flags=fl)
            e1.record(); e1.synchronize()
            print(dt, "flags", fl, "us/call", round(e0.elapsed_time(e1) / 8 * 1e3, 1), flush=True)
