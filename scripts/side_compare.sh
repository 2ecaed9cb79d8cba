# bench.py (two streams) under the sampler's knobs; usage: side_compare.sh "CHAINS:WG_PER_CU:HOT_KB:SKEW ..."
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in $1; do
IFS=: read nc wg hot skew <<< "$c"
MC_SAMPLE_CHAINS=$nc MC_SAMPLE_SKEW=$skew MC_SAMPLE_HOT_KB=$hot timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1000 --sample-wg-per-cu $wg > gpurun_out/side_x.json 2> gpurun_out/side_x.err || { tail gpurun_out/side_x.err; exit 1; }
python3 -c "
import json,sys
b=json.load(open('gpurun_out/side_x.json')); print('chains:wg:hot:skew $c', round(b['value']/1e9,3), round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['stage_ms'].items()})"; done
