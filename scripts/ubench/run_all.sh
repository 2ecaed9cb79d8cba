#!/bin/bash
# Run on the GPU box: the microbenchmarks of scripts/ubench (built in the container beforehand), each under its own time limit.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-ubench}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R/scripts/ubench" || exit 1
timeout -k 10 120 ./philox_mul > "$OUT/philox_mul.txt" 2>&1 || { echo "philox_mul failed"; exit 1; }
timeout -k 10 300 ./global_atomics 2000000 > "$OUT/global_atomics.txt" 2>&1 || { echo "global_atomics failed"; exit 1; }
timeout -k 10 120 ./gather_traffic 40 100000000 > "$OUT/gather_traffic.txt" 2>&1 || { echo "gather_traffic failed"; exit 1; }
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/gather_pmc" -o r -- "$R/scripts/ubench/gather_traffic" 40 100000000 > "$OUT/gather_pmc.txt" 2>&1 || { echo "gather pmc failed"; exit 1; }
find "$OUT/gather_pmc" -name '*counter_collection.csv' -exec cp {} "$OUT/gather_counter_collection.csv" \;
rm -rf "$OUT/gather_pmc"
echo "ubench done"
