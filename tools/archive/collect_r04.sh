#!/bin/bash
# Copy the summaries of tools/measure_r04.sh (gpurun_out/m4) and tools/measure_r04_sim.sh (gpurun_out/r4s) into profiles/ (tracked).
O=gpurun_out/m4; S=gpurun_out/r4s; P=profiles
ls -t $(find $O/prof_bench -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r04_kernel_stats.csv
ls -t $(find $O/prof_eager -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r04_kernel_stats_eager.csv
grep '^{"metric"' $O/prof_bench.json > $P/r04_kernel_stats_bench_line.json
grep '^{"metric"' $O/prof_eager.json > $P/r04_kernel_stats_eager_bench_line.json
grep '^{"metric"' $O/bench.json > $P/r04_bench.json
grep '^{"metric"' $O/bench_gloo2.json > $P/r04_bench_gloo2.json
grep '^{"metric"' $O/bench_mm0.json > $P/r04_bench_multimask0.json
cp $O/pmc_traffic.json $P/r04_pmc_traffic.json
cp $O/attn_bench.jsonl $P/r04_attention_bench.jsonl
cp $O/gemm_shapes.jsonl $P/r04_gemm_shapes.jsonl
for c in L f32 host eager; do grep '^{"metric"' $O/bench_$c.json > $P/r04_bench_$c.json; done
mv $P/r04_bench_L.json $P/r04_bench_L_config.json; mv $P/r04_bench_f32.json $P/r04_bench_fp32_exact_mode.json
mv $P/r04_bench_host.json $P/r04_bench_host_inputs.json; mv $P/r04_bench_eager.json $P/r04_bench_eager_launches.json
python3 tools/sim_show.py $S > $P/r04_similarity_summary.txt
cat $S/sim_bench.jsonl > $P/r04_similarity_bench.jsonl
cat $S/sim_bench_global.jsonl > $P/r04_similarity_bench_global_threshold.jsonl
for s in 512x12500 256x12500 512x125000 1m; do ls -t $(find $S/prof_$s -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $P/r04_similarity_kernel_stats_$s.csv; done
ls -la $P/r04_*
