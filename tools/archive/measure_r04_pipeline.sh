#!/bin/bash
# Round-4 re-measurement after bench.py's default became two forwards in flight (ForwardPipeline): one gpurun call.
# Outputs under gpurun_out/m5/ ; the summaries are copied to profiles/r04_* by hand (listed in DESIGN 5).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/m5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2>/dev/null || exit 1
echo "kernel trace (two forwards in flight) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 4 --warmup 1 --graph 0 --overlap 0 --no-cpu-baseline > $O/prof_eager.json 2>/dev/null || exit 1
echo "kernel trace (eager, one stream) done"
cd $R
timeout -k 10 200 python3 tools/attn_bench.py 32 > $O/attn_bench.jsonl 2>/dev/null || exit 1
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; grep '^{"metric"' $O/bench.json | cut -c1-200
: > $O/driver_form.jsonl
for v in "--inflight 2" "--inflight 1" "--inflight 2" "--inflight 1" "--inflight 2"; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 $v 2>/dev/null | grep '^{"metric"' >> $O/driver_form.jsonl || exit 1
done
echo "driver form done"
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 300 python3 bench.py --multimask 0 --no-cpu-baseline > $O/bench_mm0.json 2>/dev/null; echo "mm0 rc=$?"
timeout -k 10 300 python3 bench.py --sam sam_large --siglip ViT-L-16-SigLIP-384 --batch 64 --no-cpu-baseline > $O/bench_L.json 2>/dev/null; echo "L rc=$?"
timeout -k 10 300 python3 bench.py --rehearse-rccl 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl1.json 2>/dev/null; echo "rccl1 rc=$?"
