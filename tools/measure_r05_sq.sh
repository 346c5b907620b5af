#!/bin/bash
# SQ counter passes for the global attention kernel, before / after: round 4's library (tools/probes/libr04.so, built from
# commit 019c75a) against this build's default (variant 0), the fma-bias form (2) and the 64-query-per-wave form (4); separate rocprofv3 --pmc
# runs of <= 8 SQ counters, program directly after `--`. Output gpurun_out/sq5/ -> profiles/r05_pmc_sq_counters.jsonl
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE"
run() {  # tag, lib, variant
  local tag=$1 lib=$2 v=$3 i=1
  for P in "$P1" "$P2" "$P3"; do
    COR_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/${tag}_$i -- python3 $R/tools/attn_one.py $v 32 0 1 > /dev/null 2>$O/${tag}_$i.err || { echo "FAILED $tag pass $i"; tail -3 $O/${tag}_$i.err; return 1; }
    i=$((i+1))
  done
  echo "$tag done"
}
run r04_default $R/tools/probes/libr04.so 0 || exit 1
run r05_default "" 0 || exit 1
run r05_w64 "" 4 || exit 1
cd $R
for t in r04_default r05_default r05_w64; do
  for i in 1 2 3; do python3 tools/pmc_summary.py $O/${t}_$i flash_global | sed "s/^{/{\"run\": \"$t\", /"; done
done > $O/summary.jsonl
cut -c1-300 $O/summary.jsonl
