#!/bin/bash
# Same-box A/B of two builds of the library: runs <bench args> alternately with iisan_amd/libiisan_hip.so ("new") and the
# library given as $1 ("old"), three rounds.  usage (through gpurun): bash tools/ab_lib.sh tools/lib_prev.so --cached fp32 --bs 1024
old=$1; shift
cp iisan_amd/libiisan_hip.so /tmp/lib_new.so
for r in 1 2 3; do
  cp $old iisan_amd/libiisan_hip.so; echo -n "old: "; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  cp /tmp/lib_new.so iisan_amd/libiisan_hip.so; echo -n "new: "; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
