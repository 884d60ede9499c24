#!/bin/bash
# One rocprofv3 PMC pass (counters only, no traces beside --kernel-trace) of a bench.py configuration; per-kernel averages.
# usage: bash tools/pmc_run.sh <tag> "<counter list>" <kernel substring> <bench.py args...>
tag=$1; ctrs=$2; filt=$3; shift 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace -d $out -o p -- python3 bench.py "$@" > $out/bench.log 2>&1
python3 tools/pmc_summary.py $(find $out -name "*.db" | head -1) "$filt" | tee $out/summary.txt
find $out -name "*.db" -delete
