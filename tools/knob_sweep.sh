#!/bin/bash
# Same-box sweep of one dev knob: bash tools/knob_sweep.sh <knob> "<values>" <bench args>   (IISAN_DEV_KNOBS; value "-" = knob unset)
knob=$1; vals=$2; shift 2
for r in 1 2; do
  for v in $vals; do
    if [ "$v" = "-" ]; then unset IISAN_DEV_KNOBS; else export IISAN_DEV_KNOBS="$knob=$v"; fi
    echo -n "$knob=$v: "; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done
