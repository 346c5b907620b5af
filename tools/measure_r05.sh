#!/bin/bash
# Round-5 measurement pass on the GPU box (one gpurun call): bench lines (default, driver's form, every BASELINE config / rehearsal), rocprofv3
# kernel stats of the SAME command in eager single-stream mode (its sums reproduce roofline.frac) and in graph mode, PMC traffic of the GEMMs
# (separate passes; 2 eager forwards -> bytes per step + launches per step), micro-benchmarks. Outputs: gpurun_out/m5/ ; tools/collect_r05.sh
# copies the summaries to profiles/r05_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/m5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --steps 4 --warmup 1 --graph 0 --overlap 0 --no-cpu-baseline > $O/prof_eager.json 2>/dev/null || exit 1
echo "kernel trace (eager, one stream) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2>/dev/null || exit 1
echo "kernel trace (graph mode, two in flight) done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_bench_$C -- python3 $R/bench.py --steps 1 --warmup 1 --graph 0 --overlap 0 --inflight 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
done
echo "pmc done"
cd $R
python3 tools/pmc_traffic.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_traffic.json --forwards 2 || exit 1
cp $O/pmc_traffic.json $R/profiles/r05_pmc_traffic.json   # bench.py quotes it only when gemm_source_id AND launches per step match
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; grep '^{"metric"' $O/bench.json | cut -c1-160
timeout -k 10 400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2>/dev/null || exit 1
timeout -k 10 400 python3 $R/bench.py --config 1 --no-cpu-baseline > $O/bench_config1.json 2>/dev/null; echo "config1 rc=$?"
timeout -k 10 400 python3 $R/bench.py --config 3 --no-cpu-baseline > $O/bench_config3.json 2>/dev/null; echo "config3 rc=$?"
timeout -k 10 500 python3 $R/bench.py --config 2 --gpus 2 --backend gloo --no-cpu-baseline > $O/bench_config2_gloo2.json 2> $O/bench_config2_gloo2.err; echo "config2 gloo2 rc=$?"
timeout -k 10 600 python3 $R/bench.py --config 4 --gpus 4 --backend gloo --gallery 200000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config4_gloo4.json 2> $O/bench_config4_gloo4.err; echo "config4 gloo4 rc=$?"
timeout -k 10 300 python3 $R/bench.py --rehearse-rccl 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rccl1.json 2>/dev/null; echo "rccl one rank rc=$?"
timeout -k 10 200 python3 tools/attn_bench.py 32 > $O/attn_bench.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/gemm_shapes.py > $O/gemm_shapes.jsonl 2>/dev/null || exit 1
timeout -k 10 300 python3 tools/sim_bench.py > $O/sim_bench.jsonl 2>/dev/null || exit 1
echo "micro benches done"
timeout -k 10 300 python3 bench.py --dtype f32 --batch 8 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --host-inputs 1 --no-cpu-baseline > $O/bench_host.json 2>/dev/null || exit 1
echo "other configs done"
