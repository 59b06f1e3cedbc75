#!/bin/bash
# One GPU session = a list of named steps, each under its own time limit; a step that is killed,
# aborts or faults ends the session (nothing more touches the GPU after it).
#
#   bash benchmarks/gpu/run.sh <step> [<step> ...]
#
#   tests[:<pytest args>]        GPU tests (default: the whole suite)
#   smoke                        __graft_entry__.smoke()
#   bench[:<bench.py args>]      the headline benchmark
#   py:<limit_s>:<script + args> any benchmark script, output in gpurun_out/<script>.log
#   prof:<limit_s>:<name>:<script + args>   rocprofv3 kernel trace + stats of a script, summary in
#                                gpurun_out/prof_<name>_summary.txt
#   pmc:<name>:<counters>:<script + args>   one counter pass (its own run, 120 s limit)
set -u
ROOT=$PWD
mkdir -p "$ROOT/gpurun_out"
step() {  # step <limit_s> <log> <cmd...>
  local lim=$1 log=$2; shift 2
  echo "== $* (limit ${lim}s)" | tee -a "$ROOT/gpurun_out/session.log"
  timeout -k 10 "$lim" "$@" > "$ROOT/gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a "$ROOT/gpurun_out/session.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
    echo "ABORT after $log (rc=$rc)" | tee -a "$ROOT/gpurun_out/session.log"; tail -30 "$ROOT/gpurun_out/$log"; exit $rc
  fi
  return $rc
}
for spec in "$@"; do
  kind=${spec%%:*}; rest=${spec#*:}; [ "$rest" = "$spec" ] && rest=""
  case $kind in
    tests)
      step 1100 t_gpu.log python -u -m pytest -q --timeout 300 --timeout-method thread ${rest:-tests} -m gpu \
        || { grep -E "FAILED|^E  " gpurun_out/t_gpu.log | tail -40; exit 1; }
      tail -2 gpurun_out/t_gpu.log ;;
    smoke)
      step 300 smoke.log python __graft_entry__.py smoke; tail -1 gpurun_out/smoke.log ;;
    bench)
      step 400 bench.log python bench.py $rest; grep '^{' gpurun_out/bench.log | tail -1 ;;
    py)
      lim=${rest%%:*}; cmd=${rest#*:}; name=$(basename "${cmd%% *}" .py)
      step "$lim" "$name.log" python -u $cmd; tail -12 "gpurun_out/$name.log" ;;
    prof)
      lim=${rest%%:*}; rest=${rest#*:}; name=${rest%%:*}; cmd=${rest#*:}
      script=${cmd%% *}; args=${cmd#"$script"}
      cd /tmp && export TMPDIR=/tmp
      step "$lim" "prof_$name.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$name" \
        -o "$name" -- python3 "$ROOT/$script" $args
      cd "$ROOT"
      f=$(find "gpurun_out/prof_$name" -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 40 > "gpurun_out/prof_${name}_summary.txt" 2>&1
      find "gpurun_out/prof_$name" -name '*kernel_trace.csv' -delete
      head -14 "gpurun_out/prof_${name}_summary.txt" ;;
    pmc)
      name=${rest%%:*}; rest=${rest#*:}; ctr=${rest%%:*}; cmd=${rest#*:}
      script=${cmd%% *}; args=${cmd#"$script"}
      cd /tmp && export TMPDIR=/tmp
      step 120 "pmc_$name.log" timeout -s KILL 110 rocprofv3 --pmc $ctr --output-format csv -d "$ROOT/gpurun_out/pmc_$name" \
        -o "$name" -- python3 "$ROOT/$script" $args
      cd "$ROOT"
      f=$(find "gpurun_out/pmc_$name" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python benchmarks/pmc_summary.py "$f" > "gpurun_out/pmc_${name}_summary.txt" 2>&1
      head -12 "gpurun_out/pmc_${name}_summary.txt" ;;
    *) echo "unknown step $spec"; exit 2 ;;
  esac
done
