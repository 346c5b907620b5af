set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sqsim; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/sim1m_$i -- python3 $R/tools/sim_bench.py 1m 4 > /dev/null 2>$O/sim1m_$i.err || { echo FAILED $i; tail -3 $O/sim1m_$i.err; exit 1; }
  i=$((i+1))
done
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py $O/sim1m_$i "sim_scan<unsigned short, 2, false>" | cut -c1-900; done
