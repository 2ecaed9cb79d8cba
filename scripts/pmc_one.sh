#!/bin/bash
# rocprofv3 --pmc passes of one python command on the GPU box, one run per ';'-separated counter group, and the
# per-kernel means: scripts/pmc_one.sh <tag> "<SQ_... SQ_...;TCP_... TCC_...>" <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; GROUPS_S=$2; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 1
i=0
IFS=';' read -ra GROUPS_ <<< "$GROUPS_S"
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmcx_$i" -o r -- python3 "$@" > "$OUT/pmcx_$i.out" 2> "$OUT/pmcx_$i.err" \
    || { rc=$?; tail -20 "$OUT/pmcx_$i.err"; echo "FAILED group $i"; exit $rc; }
done
python3 scripts/pmcx_summary.py "$OUT" > "$OUT/pmcx_summary.txt"
find "$OUT" -name '*counter_collection.csv' -delete
cat "$OUT/pmcx_summary.txt"
