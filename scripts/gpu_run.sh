#!/bin/bash
# One GPU-box session: every GPU step under its own time limit; a step that ends by a signal, an abort or its time limit ends
# the session (nothing more is started on the GPU after a possible fault).  Usage: scripts/gpu_run.sh <step> [<step> ...]
#   steps: valu | tests [pytest args] is spelled tests:<k-expression> | bench:<config> | profile:<config> | cmd:<shell words>
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
ok_or_stop() {   # $1 = exit status, $2 = what
  case "$1" in
    0|1) return 0 ;;                       # 1 = test failures / a failed assertion: not a GPU fault
    *) echo "[gpu_run] step '$2' ended with status $1: stopping"; exit "$1" ;;
  esac
}
for step in "$@"; do
  echo "[gpu_run] === $step"; date +%T
  case "$step" in
    valu)
      timeout -k 10 120 scripts/bin/valu_peak > gpurun_out/valu_peak.jsonl 2> gpurun_out/valu_peak.err; rc=$?
      tail -30 gpurun_out/valu_peak.jsonl; ok_or_stop $rc "$step" ;;
    tests*)
      expr="${step#tests}"; expr="${expr#:}"
      if [ -n "$expr" ]; then
        timeout -k 10 1100 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -k "$expr" > gpurun_out/pytest_gpu.log 2>&1; rc=$?
      else
        timeout -k 10 1100 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
      fi
      grep -E "FAILED|ERROR|passed|failed" gpurun_out/pytest_gpu.log | tail -40; ok_or_stop $rc "$step" ;;
    bench:*)
      cfg="${step#bench:}"
      timeout -k 10 600 python -u bench.py --config "$cfg" --steps 3 --warmup 1 > "gpurun_out/bench_$cfg.json" 2> "gpurun_out/bench_$cfg.err"; rc=$?
      tail -c 3000 "gpurun_out/bench_$cfg.json"; tail -5 "gpurun_out/bench_$cfg.err"; ok_or_stop $rc "$step" ;;
    profile:*)
      cfg="${step#profile:}"
      timeout -k 10 1000 scripts/profile.sh "prof_$cfg" --config "$cfg"; rc=$?
      ok_or_stop $rc "$step" ;;
    cmd:*)
      timeout -k 10 900 bash -c "${step#cmd:}"; rc=$?
      ok_or_stop $rc "$step" ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo "[gpu_run] all steps done"; date +%T
