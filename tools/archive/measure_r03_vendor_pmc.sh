#!/bin/bash
# Counter passes (separate rocprofv3 --pmc runs) of the vendor GEMM kernel and of gemm_pp on the qkv shape, same box, for the yardstick.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/vpmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SMEM"
P3="GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/lib_$i -- python3 $R/tools/lib_gemm_one.py 131072 2304 768 > /dev/null 2>$O/lib_$i.err || { echo "FAILED lib pass $i"; tail -3 $O/lib_$i.err; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pp_$i -- python3 $R/tools/gemm_one.py 131072 2304 768 > /dev/null 2>$O/pp_$i.err || { echo "FAILED pp pass $i"; tail -3 $O/pp_$i.err; exit 1; }
  i=$((i+1))
done
cd $R
for t in lib pp; do for i in 1 2 3 4 5; do python3 tools/pmc_summary.py $O/${t}_$i Cijk gemm_pp | sed "s/^{/{\"run\": \"$t\", /"; done; done > $O/summary.jsonl
# kernel durations of the same runs
for t in lib pp; do f=$(find $O/${t}_3 -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
d = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm_pp" in n:
        d.setdefault(n[:60], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v.sort(); print('{"kernel": "%s", "median_us": %.1f, "launches": %d}' % (k, v[len(v) // 2] / 1e3, len(v)))
PY
done >> $O/summary.jsonl
cat $O/summary.jsonl | cut -c1-600
