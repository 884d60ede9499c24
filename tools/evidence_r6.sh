#!/bin/bash
# (round 6: the headline passes run with --no-overlap-towers — kernels of two overlapped towers share the CUs and their durations are not kernel measures)
# The round's evidence in one gpurun call: kernel traces, HBM-side traffic (PMC, two passes each) and matrix-pipe counters of the four
# bench configurations.  usage: bash tools/evidence_r6.sh <tag> [parts: trace pmc mfma]      -> gpurun_out/ev_<tag>/
tag=${1:-r6}; parts=${2:-"trace pmc mfma"}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_$tag
mkdir -p $out
HEAD="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-check --no-overlap-towers"
HEAD2="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check --no-overlap-towers"
C3="--cached fp32 --bs 1024 --no-cpu-baseline"
C5="--cached fp16 --versa --bs 128 --no-cpu-baseline"
EV="--eval"
db() { find $1 -name "*.db" | head -1; }
if [[ $parts == *trace* ]]; then
  for cfg in head cached versa eval; do
    case $cfg in head) args=$HEAD;; cached) args=$C3;; versa) args=$C5;; eval) args=$EV;; esac
    d=$out/trace_$cfg; rm -rf $d; mkdir -p $d
    rocprofv3 --kernel-trace --stats -d $d -o t -- python3 bench.py $args > $d/bench.log 2>&1
    python3 tools/rocpd_summary.py $(db $d) > $out/${cfg}_kernel_stats.md 2>&1
    python3 tools/rocpd_summary.py $(db $d) --by-grid > $out/${cfg}_kernel_stats_by_grid.md 2>&1
    python3 tools/rocpd_seq.py $(db $d) > $out/${cfg}_sequence.txt 2>/dev/null
    grep "^{" $d/bench.log | tail -1 > $out/${cfg}_bench_line.json
    find $d -name "*.db" -delete
  done
fi
if [[ $parts == *pmc* ]]; then
  pass() { local name=$1 ctr=$2; shift 2; local d=$out/pmc_${name}_$ctr; rm -rf $d; mkdir -p $d
           rocprofv3 --pmc $ctr --kernel-trace -d $d -o p -- python3 bench.py "$@" > $d/bench.log 2>&1; db $d; }
  f=$(pass head FETCH_SIZE $HEAD2); w=$(pass head WRITE_SIZE $HEAD2)
  python3 tools/pmc_traffic.py $f $w gemm16 > $out/pmc_head_gemm16.md 2>&1
  python3 tools/pmc_traffic.py $f $w layernorm768 > $out/pmc_head_ln.md 2>&1
  python3 tools/pmc_traffic.py $f $w stream_stats > $out/pmc_head_finalize.md 2>&1
  python3 tools/pmc_traffic.py $f $w fold_ln > $out/pmc_head_fold.md 2>&1
  python3 tools/pmc_traffic.py $f $w attention16 > $out/pmc_head_attn.md 2>&1
  f=$(pass cached FETCH_SIZE $C3 --steps 2 --warmup 1); w=$(pass cached WRITE_SIZE $C3 --steps 2 --warmup 1)
  python3 tools/pmc_traffic.py --step $f $w 5 cached_fp32_bs1024 $out/pmc_traffic_cached.json > $out/pmc_cached.md 2>&1
  f=$(pass versa FETCH_SIZE $C5 --steps 2 --warmup 1); w=$(pass versa WRITE_SIZE $C5 --steps 2 --warmup 1)
  python3 tools/pmc_traffic.py --step $f $w 5 versa_fp16_bs128 $out/pmc_traffic_cached.json > $out/pmc_versa.md 2>&1
  f=$(pass eval FETCH_SIZE $EV); w=$(pass eval WRITE_SIZE $EV)
  python3 tools/pmc_traffic.py $f $w score_rank > $out/pmc_eval_score_rank.md 2>&1
  python3 tools/pmc_traffic.py $f $w score_topk > $out/pmc_eval_score_topk.md 2>&1
  python3 tools/pmc_traffic.py $f $w gemm32 > $out/pmc_eval_gemm32.md 2>&1
  find $out -name "*.db" -delete
fi
if [[ $parts == *mfma* ]]; then
  CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  for cfg in head cached eval; do
    case $cfg in head) args=$HEAD2;; cached) args="$C3 --steps 2 --warmup 1";; eval) args=$EV;; esac
    d=$out/mfma_$cfg; rm -rf $d; mkdir -p $d
    rocprofv3 --pmc $CTR --kernel-trace -d $d -o p -- python3 bench.py $args > $d/bench.log 2>&1
    case $cfg in head) flt="gemm16 attention16";; cached) flt="gemm32 gemm16 ce_ sasrec_fused";; eval) flt="score_rank score_topk topk_merge gemm32";; esac
    python3 tools/mfma_util.py $(db $d) $flt > $out/mfma_$cfg.md 2>&1
    find $d -name "*.db" -delete
  done
fi
ls $out | head -50
