cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
b() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r03y_bench_$name.json 2> gpurun_out/r03y_bench_$name.err || echo "bench $name FAILED rc=$?"; }
b unlogged_a --no-cpu-baseline
b logged_a --no-cpu-baseline --log
b unlogged_b --no-cpu-baseline
b logged_b --no-cpu-baseline --log
b logged200 --no-cpu-baseline --log --steps 200
b unlogged200 --no-cpu-baseline --steps 200
for f in gpurun_out/r03y_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f'.split('bench_')[1], round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| step', round(d['roofline']['launch_ms']*1e3,1), 'ckpt ms', c.get('end_of_learn_checkpoint_ms'))
" 2>&1 | tail -1; done
