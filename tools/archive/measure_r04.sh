#!/bin/bash
# Round-4 measurement pass on the GPU box (one gpurun call): the bench line, rocprofv3 kernel stats of the SAME command in graph mode and
# in eager single-stream mode (the summary of the eager pass reproduces that run's roofline.frac), PMC traffic of the
# GEMMs (separate passes), the 2-rank gloo rehearsal started by bench.py itself, multimask_output=False, other configurations, GEMM /
# attention micro-benchmarks. Outputs under gpurun_out/m4/ ; tools/collect_r04.sh copies the summaries to profiles/r04_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/m4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2>/dev/null || exit 1
echo "kernel trace (graph mode) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 4 --warmup 1 --graph 0 --overlap 0 --no-cpu-baseline > $O/prof_eager.json 2>/dev/null || exit 1
echo "kernel trace (eager, one stream) done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_bench_$C -- python3 $R/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline > /dev/null 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bench_GRBM -- python3 $R/bench.py --steps 2 --warmup 1 --graph 0 --overlap 0 --no-cpu-baseline > /dev/null 2>&1 || exit 1
echo "pmc done"
cd $R
python3 tools/pmc_traffic.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_traffic.json > /dev/null || exit 1
cp $O/pmc_traffic.json $R/profiles/r04_pmc_traffic.json   # bench.py quotes it only when its gemm_source_id matches this build
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; grep '^{"metric"' $O/bench.json | cut -c1-200
timeout -k 10 400 python3 $R/bench.py --gpus 2 --backend gloo --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 300 python3 $R/bench.py --multimask 0 --no-cpu-baseline > $O/bench_mm0.json 2>/dev/null; echo "mm0 rc=$?"
timeout -k 10 200 python3 tools/attn_bench.py 32 > $O/attn_bench.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/gemm_shapes.py > $O/gemm_shapes.jsonl 2>/dev/null || exit 1
echo "micro benches done"
timeout -k 10 300 python3 bench.py --sam sam_large --siglip ViT-L-16-SigLIP-384 --batch 64 --no-cpu-baseline > $O/bench_L.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --dtype f32 --batch 8 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --host-inputs 1 --no-cpu-baseline > $O/bench_host.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --graph 0 --no-cpu-baseline > $O/bench_eager.json 2>/dev/null || exit 1
echo "other configs done"
