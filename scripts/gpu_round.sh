#!/bin/bash
# Run on the GPU box (through gpurun): GPU tests, bench, rocprofv3 kernel trace and the two
# HBM-traffic counter passes, the VALU counter passes, a kernel trace of the other configs' legs.
# Usage: scripts/gpu_round.sh <tag> [tests|bench|prof|pmc|valu|others ...]
# Every GPU step has its own time limit and the script stops at the first failure.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
shift || true
STEPS=${*:-tests bench prof pmc valu others}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1

die() { echo "FAILED: $1 (rc=$2)"; exit "$2"; }

for s in $STEPS; do
  case $s in
    tests)
      timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread \
        > "$OUT/gpu_tests.log" 2>&1 || { rc=$?; tail -30 "$OUT/gpu_tests.log"; die tests $rc; }
      tail -3 "$OUT/gpu_tests.log" ;;
    bench)
      timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" \
        || { rc=$?; tail -20 "$OUT/bench.err"; die bench $rc; }
      cat "$OUT/bench.json" ;;
    prof)
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- \
        python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs > "$OUT/bench_prof.json" 2> "$OUT/prof.err" \
        || { rc=$?; tail -20 "$OUT/prof.err"; die prof $rc; }
      find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
      rm -f $(find "$OUT/trace" -name '*kernel_trace.csv')   # large; the stats are what we keep
      head -8 "$OUT/kernel_stats.csv" | cut -c1-160 ;;
    pmc)   # (--serial: the summary tells the three sorts of a step apart by the order of the dispatches)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o r -- \
          python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --serial --sample-wg-per-cu 7 > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" \
          || { rc=$?; tail -20 "$OUT/pmc_$c.err"; die pmc_$c $rc; }
      done
      python3 scripts/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json" && cat "$OUT/pmc_summary.json" ;;
    valu)   # k_sample's vector-ALU counters against a saturating yardstick (profiles/<name>_valu.json)
      bash scripts/valu_pass.sh "$TAG" > "$OUT/valu_pass.log" 2>&1 || { rc=$?; tail -20 "$OUT/valu_pass.log"; die valu $rc; }
      tail -3 "$OUT/valu_pass.log" ;;
    others)   # kernel trace of the legs behind the timed region (configs 3, 4, 5, the sub-word scorer)
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_o" -o r -- \
        python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_others.json" 2> "$OUT/others.err" \
        || { rc=$?; tail -20 "$OUT/others.err"; die others $rc; }
      find "$OUT/trace_o" -name '*kernel_stats.csv' -exec cp {} "$OUT/other_configs_kernel_stats.csv" \;
      rm -rf "$OUT/trace_o"
      head -12 "$OUT/other_configs_kernel_stats.csv" | cut -c1-140 ;;
    pmcx)   # free-form counter passes: PMCX="SQ_WAVES SQ_INSTS_VALU;TCP_..." (one rocprofv3 run per ';' group)
      i=0
      IFS=';' read -ra GROUPS_ <<< "${PMCX:-SQ_WAVES}"
      for grp in "${GROUPS_[@]}"; do
        i=$((i+1))
        timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmcx_$i" -o r -- \
          python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --serial --sample-wg-per-cu 7 > "$OUT/pmcx_$i.json" 2> "$OUT/pmcx_$i.err" \
          || { rc=$?; tail -20 "$OUT/pmcx_$i.err"; die pmcx_$i $rc; }
      done
      python3 scripts/pmcx_summary.py "$OUT" | tee "$OUT/pmcx_summary.txt" ;;
  esac
done
echo "gpu_round $TAG done"
