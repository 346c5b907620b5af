#!/bin/bash
# Round-3 measurement pass on the GPU box (one gpurun call): bench, rocprof kernel stats, PMC traffic (separate passes), SQ counters,
# similarity / attention / GEMM micro-benchmarks, other configurations, race screen. Outputs under gpurun_out/m3/ ; the summaries are
# copied to profiles/r03_* afterwards (tools/collect_r03.sh).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/m3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2>/dev/null || exit 1
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_bench_$C -- python3 $R/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_sim_$C -- python3 $R/tools/sim_bench.py 1m 4 > /dev/null 2>&1 || exit 1
done
echo "pmc done"
cd $R
python3 tools/pmc_traffic.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_traffic.json > /dev/null || exit 1
cp $O/pmc_traffic.json $R/profiles/r03_pmc_traffic.json   # bench.py quotes it only when its gemm_source_id matches this build
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; grep '^{"metric"' $O/bench.json | cut -c1-200
python3 tools/pmc_traffic.py $O/pmc_sim_FETCH_SIZE $O/pmc_sim_WRITE_SIZE $O/sim_pmc_traffic.json "sim_scan<unsigned short, 2, false>" > /dev/null || exit 1
timeout -k 10 200 python3 tools/sim_bench.py > $O/sim_bench.jsonl 2>/dev/null || exit 1
cd /tmp; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sim -- python3 $R/tools/sim_bench.py 1m 6 > /dev/null 2>&1 || exit 1; cd $R
timeout -k 10 200 python3 tools/attn_bench.py 32 > $O/attn_bench.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/gemm_shapes.py > $O/gemm_shapes.jsonl 2>/dev/null || exit 1
echo "micro benches done"
timeout -k 10 300 python3 bench.py --sam sam_large --siglip ViT-L-16-SigLIP-384 --batch 64 --no-cpu-baseline > $O/bench_L.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --dtype f32 --batch 8 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --host-inputs 1 --no-cpu-baseline > $O/bench_host.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --graph 0 --no-cpu-baseline > $O/bench_eager.json 2>/dev/null || exit 1
echo "other configs done"
bash tools/measure_r03_sq.sh > $O/sq.log 2>&1; echo "sq rc=$?"
bash tools/measure_r03_gemm_pmc.sh 0 > $O/gemm_pmc.log 2>&1; echo "gemm pmc rc=$?"
timeout -k 10 400 python3 tools/race_screen.py 20 > $O/race_screen.txt 2>&1; echo "race rc=$?"; tail -2 $O/race_screen.txt
