# round 3, GPU call 1: the new tests, then the bench lines the round-2 review asked for (shard sizes, world-2 on one device, logged loop)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -rs --durations=8 > gpurun_out/r03a_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_gputest.log
tail -25 gpurun_out/r03a_gputest.log
b() { name=$1; shift; timeout 300 python bench.py "$@" > gpurun_out/r03a_bench_$name.json 2> gpurun_out/r03a_bench_$name.err || echo "bench $name FAILED rc=$?"; }
b default
b 2048 --no-cpu-baseline --envs-per-gpu 2048
b 1024 --no-cpu-baseline --envs-per-gpu 1024
b logged --no-cpu-baseline --log
b gloo2_same_device --no-cpu-baseline --gpus 2 --backend gloo --same-device --steps 20
b rccl_1rank --no-cpu-baseline --force-dist --steps 20
for f in gpurun_out/r03a_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
c=d['config']
print(' ', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| n_gpus', d['n_gpus'], c.get('backend'), 'allreduce_us', c.get('grad_allreduce_us'), '| step', round(d['roofline']['launch_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],4), '| update', round(d.get('roofline_update',{}).get('launch_ms',0)*1e3,1), 'us frac', round(d.get('roofline_update',{}).get('frac',0),4))
" 2>&1 | tail -2; done
tail -4 gpurun_out/r03a_bench_*.err | grep -v "amdgpu.ids" | tail -30
