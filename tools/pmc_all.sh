#!/bin/bash
# All rocprofv3 PMC traffic passes of a round (counters only beside --kernel-trace; FETCH_SIZE and WRITE_SIZE need separate
# passes on gfx950).  usage (through gpurun): bash tools/pmc_all.sh <tag>      -> gpurun_out/pmc_<tag>/{head,cached,versa}.md + *.json
tag=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
pass() {   # name counter bench-args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace -d $out/${name}_$ctr -o p -- python3 bench.py "$@" > $out/${name}_$ctr.log 2>&1
  find $out/${name}_$ctr -name "*.db" | head -1
}
HEAD="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check"
f=$(pass head FETCH_SIZE $HEAD); w=$(pass head WRITE_SIZE $HEAD)
python3 tools/pmc_traffic.py $f $w gemm16 > $out/head.md 2>&1
python3 tools/pmc_traffic.py $f $w layernorm768 > $out/head_ln.md 2>&1
C3="--cached fp32 --bs 1024 --steps 2 --warmup 1 --no-cpu-baseline"
f=$(pass cached FETCH_SIZE $C3); w=$(pass cached WRITE_SIZE $C3)
python3 tools/pmc_traffic.py --step $f $w 5 cached_fp32_bs1024 $out/pmc_traffic_cached.json > $out/cached.md 2>&1
C5="--cached fp16 --versa --bs 128 --steps 2 --warmup 1 --no-cpu-baseline"
f=$(pass versa FETCH_SIZE $C5); w=$(pass versa WRITE_SIZE $C5)
python3 tools/pmc_traffic.py --step $f $w 5 versa_fp16_bs128 $out/pmc_traffic_cached.json > $out/versa.md 2>&1
find $out -name "*.db" -delete
tail -3 $out/head.md; tail -2 $out/cached.md; tail -2 $out/versa.md
