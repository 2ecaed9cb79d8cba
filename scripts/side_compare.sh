# bench.py --second-stream under the sampler's knobs; usage: side_compare.sh "HOT_KB:WG_PER_CU:SKEW:SIDE_PRIORITY ..."
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in $1; do
IFS=: read hot wg skew prio <<< "$c"
BENCH_SIDE_PRIORITY=$prio MC_SAMPLE_SKEW=$skew MC_SAMPLE_HOT_KB=$hot MC_SAMPLE_WG_PER_CU=$wg timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 1000 --second-stream > gpurun_out/side_x.json 2> gpurun_out/side_x.err || { tail gpurun_out/side_x.err; exit 1; }
python3 -c "
import json,sys
b=json.load(open('gpurun_out/side_x.json')); print('hot:wg:skew:prio $c', round(b['value']/1e9,3), round(b['ms_per_step'],4), {k:round(v,3) for k,v in b['stage_ms'].items()})"; done
