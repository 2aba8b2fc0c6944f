# round 6: the bench lines DESIGN.md / profiles/ quote (-> gpurun_out/r06z_bench_*.json; copy to profiles/r06_bench_*.json). GPU box, repo root.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
b() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/r06z_bench_$name.json 2> gpurun_out/r06z_bench_$name.err || { echo "bench $name FAILED rc=$?"; tail -5 gpurun_out/r06z_bench_$name.err; }; }
b default
b driver_style --steps 20 --warmup 5
b standing --no-cpu-baseline --regime standing
b iters2 --no-cpu-baseline --contact-iters 2
b 2048 --no-cpu-baseline --envs-per-gpu 2048
b 1024 --no-cpu-baseline --envs-per-gpu 1024
b cfg2_8192_grid --no-cpu-baseline --envs-per-gpu 8192 --terrain grid --steps 30
b 16384_flat --no-cpu-baseline --envs-per-gpu 16384 --steps 20
b dagger_every_1 --no-cpu-baseline --dagger-every 1 --steps 30
b gloo2_same_device --no-cpu-baseline --gpus 2 --backend gloo --same-device --steps 20
b gloo8_same_device_global16384 --no-cpu-baseline --gpus 8 --backend gloo --same-device --global-envs 16384 --steps 10
b gloo8_same_device_global32768_dagger1 --no-cpu-baseline --gpus 8 --backend gloo --same-device --global-envs 32768 --dagger-every 1 --steps 6 --warmup 2
b rccl_1rank --no-cpu-baseline --force-dist --steps 30
b logged_200 --no-cpu-baseline --log --steps 200
for f in gpurun_out/r06z_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f'.split('bench_')[1], round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| step', round(d['roofline']['launch_ms']*1e3,1), 'frac', round(d['roofline']['frac'],4), '| upd', round(d['roofline_update']['launch_ms']*1e3,1), round(d['roofline_update']['frac'],4), '| ar', c.get('grad_allreduce_us'), '| dagger it', c.get('dagger_iterations'))
" 2>&1 | tail -1; done
