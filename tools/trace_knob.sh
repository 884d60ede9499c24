#!/bin/bash
# kernel trace of one bench configuration under a dev-knob setting: bash tools/trace_knob.sh <outdir> "<IISAN_DEV_KNOBS or ->" <bench args>
out=$1; knobs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$knobs" != "-" ]; then export IISAN_DEV_KNOBS="$knobs"; fi
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o t -- python3 bench.py "$@" --no-cpu-baseline > $out/bench.log 2>&1
db=$(find $out -name "*.db" | head -1)
python3 tools/rocpd_summary.py $db > $out/kernel_stats.md 2>&1
python3 tools/rocpd_seq.py $db > $out/sequence.txt 2>/dev/null
find $out -name "*.db" -delete
grep -o '"ms_per_step": [0-9.]*' $out/bench.log | head -1
