cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_env_runner.py tests/test_gpu_policy_kernel.py tests/test_wg_golden.py -m gpu -q -x --timeout 180 2>&1 | tail -25
b() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r03x_bench_$name.json 2> gpurun_out/r03x_bench_$name.err || { echo "bench $name FAILED rc=$?"; tail -5 gpurun_out/r03x_bench_$name.err; }; }
b unlogged_a --no-cpu-baseline
b logged_a --no-cpu-baseline --log
b unlogged_b --no-cpu-baseline
for f in gpurun_out/r03x_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f'.split('bench_')[1], round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| step', round(d['roofline']['launch_ms']*1e3,1))
" 2>&1 | tail -1; done
