#!/bin/bash
# Largest configuration: sam_huge + ViT-SO400M-14-SigLIP-384 (the factory's default tower), batch 16: bench line + rocprofv3 kernel stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/H; mkdir -p $O
cd $R
timeout -k 10 500 python3 bench.py --sam sam_huge --siglip ViT-SO400M-14-SigLIP-384 --batch 16 --no-cpu-baseline > $O/bench_H.json 2> $O/bench_H.err || { tail -5 $O/bench_H.err; exit 1; }
cut -c1-300 $O/bench_H.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --sam sam_huge --siglip ViT-SO400M-14-SigLIP-384 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2>/dev/null || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_H.csv
head -12 $O/kernel_stats_H.csv | cut -c1-200
