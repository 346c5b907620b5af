#!/bin/bash
# SQ counter passes (separate rocprofv3 --pmc runs, <= 8 SQ counters each, program directly after `--`) for the kernels that are
# ~80 % of a step: gemm_pp (qkv and lin1+GELU shapes of the batch-32 SAM-B block), flash_global_pipe, win_attn.
# Output: gpurun_out/sq/<tag>_<pass>/ ; summarise with tools/pmc_summary.py -> profiles/r03_pmc_sq_counters.jsonl
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE"
run() {  # tag, program args...
  local tag=$1; shift
  local i=1
  for P in "$P1" "$P2" "$P3"; do
    timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/${tag}_$i -- python3 "$@" > /dev/null 2>$O/${tag}_$i.err || { echo "FAILED $tag pass $i"; tail -3 $O/${tag}_$i.err; return 1; }
    i=$((i+1))
  done
  echo "$tag done"
}
run gemm_qkv $R/tools/gemm_one.py 131072 2304 768 || exit 1
run gemm_lin1 $R/tools/gemm_one.py 131072 3072 768 0 gelu || exit 1
run gemm_proj $R/tools/gemm_one.py 131072 768 768 0 res || exit 1
run attn_glob $R/tools/attn_one.py 0 32 0 || exit 1
run attn_win $R/tools/attn_one.py 0 32 14 || exit 1
cd $R
for t in gemm_qkv gemm_lin1 gemm_proj attn_glob attn_win; do
  for i in 1 2 3; do python3 tools/pmc_summary.py $O/${t}_$i gemm_pp flash_global win_attn | sed "s/^{/{\"run\": \"$t\", /"; done
done > $O/summary.jsonl
cat $O/summary.jsonl | cut -c1-400
