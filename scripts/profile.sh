#!/bin/bash
# rocprofv3 evidence for a bench configuration.  Usage: scripts/profile.sh <out_dir_under_gpurun_out> [bench args...]
# One --kernel-trace --stats run, then one --pmc run per counter group (the tool does not split counters over passes); every run
# is the same command under its own time limit.  scripts/prof_summary.py condenses the output into profiles/.
set -o pipefail
OUT=gpurun_out/$1; shift
ARGS="--steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-extras $*"
mkdir -p $OUT
export TMPDIR=/tmp
echo "$ARGS" > $OUT/bench_args.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU2" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$tag -- python3 bench.py $ARGS > $OUT/pmc_$tag.log 2>&1 || { echo "pmc pass $tag failed"; tail -5 $OUT/pmc_$tag.log; exit 1; }
done
sha256sum ray-tracer_amd/lib/librtx_hip.so | cut -d' ' -f1 > $OUT/lib_sha256.txt
python3 scripts/prof_summary.py $OUT $OUT/summary.json && echo profile_done
