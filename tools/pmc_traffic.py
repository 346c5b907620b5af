#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of `python bench.py --steps 1 --warmup 1
--no-cpu-baseline`) into per-launch HBM traffic of the dominant kernel (the bf16 MFMA GEMM), following
/opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly HALF the bytes
of a wide coalesced stream (16 B per lane, global_load and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B
streaming stores. FETCH_SIZE counts L2 fabric requests, so Infinity-Cache hits are included (not pure HBM).
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--forwards N] [kernel-name substring ...]
--forwards N: the profiled command ran N forwards (bench.py --graph 0 --inflight 1 --steps 1 --warmup 1 runs 2): the file then also
carries bytes_per_step and launches_per_step, the figures bench.py quotes (and refuses when its own launch count differs).
With substrings (e.g. `sim_scan`) the same sums are taken over the kernels whose names contain any of them (similarity scan:
passes of `python tools/sim_bench.py 1m 4`)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gemm_source_id

ARGS = sys.argv[4:]
FORWARDS = None
if "--forwards" in ARGS:
    i = ARGS.index("--forwards"); FORWARDS = int(ARGS[i + 1]); del ARGS[i:i + 2]
SUBS = ARGS or ["gemm_tile<unsigned short", "gemm_pp<"]


def load(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(x in r["Kernel_Name"] for x in SUBS):
            tot += float(r["Counter_Value"]); n += 1
    return tot, n

fetch, n1 = load(sys.argv[1], "FETCH_SIZE")
write, n2 = load(sys.argv[2], "WRITE_SIZE")
assert n1 == n2 and n1 > 0
out = dict(kernel="bf16-operand GEMM kernels (gemm_pp<*>, gemm_tile<bf16,*>)" if not ARGS else " | ".join(SUBS), launches=n1, fetch_size_kib=fetch, write_size_kib=write,
           bytes_per_launch=(2.0 * fetch + write) * 1024.0 / n1, gemm_source_id=gemm_source_id(),
           note="(2*FETCH_SIZE + WRITE_SIZE)*1024/launches; FETCH doubled per the gfx950 correction; includes Infinity-Cache hits")
if FORWARDS:
    assert n1 % FORWARDS == 0, (n1, FORWARDS)
    out.update(forwards=FORWARDS, launches_per_step=n1 // FORWARDS, bytes_per_step=(2.0 * fetch + write) * 1024.0 / FORWARDS)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
