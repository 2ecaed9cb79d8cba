#!/bin/bash
# On the GPU box: the VALU counters of k_sample and, for scale, of a kernel that keeps every SIMD's vector ALU busy
# (scripts/ubench/philox_mul: dependent Philox blocks, 1..8 waves a SIMD).  One rocprofv3 --pmc run each (8 SQ + 1 GRBM
# counters fit one pass); scripts/valu_summary.py turns them into profiles/<name>_valu.json.  Usage: valu_pass.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-valu}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT/valu_ubench" -o r -- "$R/scripts/ubench/philox_mul" \
  > "$OUT/valu_ubench.txt" 2>&1 || { rc=$?; tail -20 "$OUT/valu_ubench.txt"; echo "FAILED ubench pass"; exit $rc; }
timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/valu_bench" -o r -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --serial --sample-wg-per-cu 7 \
  > "$OUT/valu_bench.json" 2> "$OUT/valu_bench.err" || { rc=$?; tail -20 "$OUT/valu_bench.err"; echo "FAILED bench pass"; exit $rc; }
python3 scripts/valu_summary.py "$OUT" > "$OUT/valu_summary.json" || exit 1
find "$OUT" -name '*counter_collection.csv' -delete
cat "$OUT/valu_summary.json"
