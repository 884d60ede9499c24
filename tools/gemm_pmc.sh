#!/bin/bash
# L2 / L1 counters of one encoder GEMM shape (separate passes, counters only beside --kernel-trace).  usage: bash tools/gemm_pmc.sh <shape> <variant>
shape=$1; var=$2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/gemm_pmc_${shape}_$var
rm -rf $out && mkdir -p $out
i=0
for ctrs in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_IB_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace -d $out/p$i -o p -- python3 tools/gemm_pmc_probe.py $shape $var > $out/p$i.log 2>&1
  python3 tools/pmc_summary.py $(find $out/p$i -name "*.db" | head -1) gemm16 | sed 's/void (anonymous namespace):://' | awk '{print $1, $(NF-2), $(NF-1)}' 
done
find $out -name "*.db" -delete
