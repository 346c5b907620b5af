#!/bin/bash
# Round-4 similarity measurement pass (one gpurun call): parity tests of the retrieval kernels, back-to-back times of every path, per-kernel
# rocprofv3 summaries of the shard shapes. Outputs under gpurun_out/r4s/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q -k "similarity or sharded or topk" > $O/pytest_sim.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sim.log
timeout -k 10 200 python3 tools/sim_bench.py > $O/sim_bench.jsonl 2> $O/sim_bench.err || exit 1
timeout -k 10 200 python3 tools/sim_bench.py all 8 8 > $O/sim_bench_global.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/sim_bench.py all 8 24 > $O/sim_bench_global_wavefinal.jsonl 2>/dev/null || exit 1
echo "bench done"
cd /tmp; export TMPDIR=/tmp
for S in 512x12500 256x12500 512x32000 32x100000 512x125000 1m; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$S -- python3 $R/tools/sim_bench.py $S 12 > /dev/null 2>&1 || exit 1
  f=$(ls -t $(find $O/prof_$S -name "*kernel_stats.csv") | head -1); grep -E "sim_|Name" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-150 > $O/stats_$S.txt
done
echo "profiles done"
cd $R
