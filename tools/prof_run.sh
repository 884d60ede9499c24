#!/bin/bash
# rocprofv3 kernel trace of one bench.py configuration on the GPU box, summarised from the rocpd database.
# usage: bash tools/prof_run.sh <tag> <rows> <bench.py args...>     (run through gpurun; output under gpurun_out/prof_<tag>/)
tag=$1; rows=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o bench -- python3 bench.py "$@" > $out/bench.log 2>&1
python3 tools/rocpd_summary.py $(find $out -name "*.db" | head -1) > $out/summary.md 2>/dev/null
python3 tools/rocpd_seq.py $(find $out -name "*.db" | head -1) > $out/sequence.txt 2>/dev/null
python3 tools/rocpd_summary.py $(find $out -name "*.db" | head -1) --schema > $out/schema.txt 2>&1
python3 tools/rocpd_summary.py $(find $out -name "*.db" | head -1) --by-stream > $out/by_stream.md 2>&1
find $out -name "*.db" -delete
head -$rows $out/summary.md | cut -c1-200
grep "^{" $out/bench.log | tail -1 | cut -c1-400
