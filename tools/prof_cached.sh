cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pc && mkdir -p gpurun_out/pc
rocprofv3 --kernel-trace --stats -d gpurun_out/pc -o bench -- python3 bench.py --cached fp32 --bs 1024 --no-cpu-baseline $PROF_EXTRA > gpurun_out/pc/bench.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/pc -name "*.db" | head -1) 2>/dev/null | head -${PROF_ROWS:-24} | cut -c1-180
find gpurun_out/pc -name "*.db" -delete
grep "^{" gpurun_out/pc/bench.log | tail -1 | cut -c1-260
