#!/bin/bash
# Copy the summaries of tools/measure_r05.sh (gpurun_out/m5) and tools/measure_r05_sq.sh (gpurun_out/sq5) into profiles/ (tracked).
O=gpurun_out/m5; P=profiles
ls -t $(find $O/prof_bench -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r05_kernel_stats.csv
ls -t $(find $O/prof_eager -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r05_kernel_stats_eager.csv
grep '^{"metric"' $O/prof_bench.json > $P/r05_kernel_stats_bench_line.json
grep '^{"metric"' $O/prof_eager.json > $P/r05_kernel_stats_eager_bench_line.json
grep '^{"metric"' $O/bench.json > $P/r05_bench.json
grep '^{"metric"' $O/bench_driver.json > $P/r05_bench_driver_form.json
grep '^{"metric"' $O/bench_config1.json > $P/r05_bench_config1.json
grep '^{"metric"' $O/bench_config3.json > $P/r05_bench_config3.json
grep '^{"metric"' $O/bench_config2_gloo2.json > $P/r05_bench_config2_gloo2_rehearsal.json
grep '^{"metric"' $O/bench_config4_gloo4.json > $P/r05_bench_config4_gloo4_rehearsal.json
grep '^{"metric"' $O/bench_rccl1.json > $P/r05_bench_rccl_one_rank.json
grep '^{"metric"' $O/bench_f32.json > $P/r05_bench_fp32_exact_mode.json
grep '^{"metric"' $O/bench_host.json > $P/r05_bench_host_inputs.json
cp $O/pmc_traffic.json $P/r05_pmc_traffic.json
cp $O/attn_bench.jsonl $P/r05_attention_bench.jsonl
cp $O/gemm_shapes.jsonl $P/r05_gemm_shapes.jsonl
cp $O/sim_bench.jsonl $P/r05_similarity_bench.jsonl
[ -f gpurun_out/sq5/summary.jsonl ] && cp gpurun_out/sq5/summary.jsonl $P/r05_pmc_sq_counters.jsonl
ls $P/r05_* | wc -l
