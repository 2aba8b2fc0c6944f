cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_env_runner.py -m gpu -q -x 2>&1 | tail -3
b() { name=$1; shift; timeout 300 python bench.py "$@" > gpurun_out/r03b_bench_$name.json 2> gpurun_out/r03b_bench_$name.err || echo "bench $name FAILED rc=$?"; }
b unlogged1 --no-cpu-baseline
b logged1 --no-cpu-baseline --log
b unlogged2 --no-cpu-baseline
b logged2 --no-cpu-baseline --log
for f in gpurun_out/r03b_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f', round(d['value']), round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), 'step us', round(d['roofline']['launch_ms']*1e3,1))
"; done
timeout 300 python tools/dist_overhead.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee gpurun_out/r03b_dist_overhead.txt
timeout 900 python tools/train_walk.py 3000 survive=2.0 z=0.25 out=gpurun_out/r03b_walk_a.pt > gpurun_out/r03b_walk_a.jsonl 2> gpurun_out/r03b_walk_a.err; tail -4 gpurun_out/r03b_walk_a.jsonl
