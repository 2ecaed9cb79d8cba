#!/bin/bash
# rocprofv3 kernel trace of one python command on the GPU box: scripts/prof_one.sh <tag> <script> [args...]; prints the stats head
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python3 "$@" > "$OUT/out.txt" 2> "$OUT/err.txt" || { rc=$?; tail -20 "$OUT/err.txt"; exit $rc; }
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace"
cut -d, -f1-8 "$OUT/kernel_stats.csv" | head -${HEAD:-14}
