#!/bin/bash
# L2 / L1 counters of the QKV and FC1 products under the row-major tile list and under the panel walk (profiles/r4_gemm_l2.md).
# usage (through gpurun): bash tools/gemm_l2_pmc.sh "3:16"
walk=${1:-3:16}
for shape in qkv fc1; do
  echo "== $shape row-major list"
  GEMM_WALK="0:0" bash tools/gemm_pmc.sh $shape 4
  echo "== $shape panel walk $walk"
  GEMM_WALK="$walk" bash tools/gemm_pmc.sh $shape 4
done
