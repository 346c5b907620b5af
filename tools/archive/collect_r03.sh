#!/bin/bash
# Copy the summaries of tools/measure_r03.sh (gpurun_out/m3, scratch) into profiles/ (tracked).
O=gpurun_out/m3; P=profiles
ls -t $(find $O/prof_bench -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r03_kernel_stats.csv   # newest: gpurun_out/ keeps the files of earlier passes
grep '^{"metric"' $O/prof_bench.json > $P/r03_kernel_stats_bench_line.json
grep '^{"metric"' $O/bench.json > $P/r03_bench.json
cp $O/pmc_traffic.json $P/r03_pmc_traffic.json
cp $O/sim_pmc_traffic.json $P/r03_similarity_pmc_traffic.json
cp $O/sim_bench.jsonl $P/r03_similarity_bench.jsonl
ls -t $(find $O/prof_sim -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r03_similarity_kernel_stats.csv
cp $O/attn_bench.jsonl $P/r03_attention_bench.jsonl
cp $O/gemm_shapes.jsonl $P/r03_gemm_shapes.jsonl
for c in L f32 host eager; do grep '^{"metric"' $O/bench_$c.json > $P/r03_bench_$c.json; done
mv $P/r03_bench_L.json $P/r03_bench_L_config.json; mv $P/r03_bench_f32.json $P/r03_bench_fp32_exact_mode.json
mv $P/r03_bench_host.json $P/r03_bench_host_inputs.json; mv $P/r03_bench_eager.json $P/r03_bench_eager_launches.json
cp gpurun_out/sq/summary.jsonl $P/r03_pmc_sq_counters.jsonl
cat gpurun_out/gpmc/summary_0.txt > $P/r03_gemm_pmc_by_shape.jsonl
cp $O/race_screen.txt $P/r03_race_screen_20.txt
ls -la $P/r03_*
