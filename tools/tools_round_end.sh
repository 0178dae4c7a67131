#!/bin/bash
# one gpurun call at the end of a round: the GPU tests, the driver's bench command, the band figures.
# usage: bash tools/tools_round_end.sh <dir under gpurun_out>
R=/root/repo/gpurun_out/$1; mkdir -p $R; cd /root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $R/pytest_gpu.log 2>&1; echo "rc=$?" >> $R/pytest_gpu.log; tail -3 $R/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
for n in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $R/bench_default_$n.err | tail -1 > $R/bench_default_driver_args_1gpu_$n.json; python3 -c "
import json,sys
d=json.load(open('$R/bench_default_driver_args_1gpu_$n.json'))
print('c3', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), {k: round(v['ms_per_step'],4) for k,v in d['also'].items()})"; done
python3 tools/tools_band_time.py --workload c4 --splits 1,2,4,8 --steps 20 --out $R/band_time_c4.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 bands', [(b['split'], round(b['ms_per_step'],4)) for b in d['bands']])"
python3 tools/tools_band_time.py --workload c3 --splits 1,2,4,8 --steps 20 --out $R/band_time_c3.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 bands', [(b['split'], round(b['ms_per_step'],4)) for b in d['bands']])"
bash tools/tools_band_matrix.sh 2 "base GCM_PE_K1_SPLIT=0 GCM_PE_STOP_EVENTS=0" 0 40 80 > $R/band_matrix.txt 2>&1; cat $R/band_matrix.txt
bash tools/tools_band_trace2.sh end c4 8 > $R/c4_n8_band_stage_times.txt 2>&1; cat $R/c4_n8_band_stage_times.txt
bash tools/tools_band_trace.sh end2 c4 8 > $R/c4_n8_band_timeline.txt 2>&1
GCM_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 2 2> $R/bench_gpus2_gloo.err | grep '^{"metric' > $R/bench_gpus2_gloo_rehearsal.json; echo "gloo rehearsal rc=$?"
rm -rf /root/repo/gpurun_out/bandtrace_end*/*/*_kernel_trace.csv
