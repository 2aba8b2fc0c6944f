# round 3: the measurements DESIGN.md / profiles/ quote (bench lines -> profiles/r03_bench_*.json, kernel statistics, process-group
# overhead). usage (GPU box, repo root): bash tools/r03_final.sh ; counters and phase stamps: tools/pmc_step.sh, tools/pmc_traffic.sh,
# WBC_STAMPS=1 python tools/time_step.py 4096 200 base (after python tools/build_variant.py timing -DWBC_STEP_TIMING)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rs > gpurun_out/r03z_gputest.log 2>&1; grep -a "passed\|failed\|SKIPPED" gpurun_out/r03z_gputest.log
b() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r03z_bench_$name.json 2> gpurun_out/r03z_bench_$name.err || { echo "bench $name FAILED rc=$?"; tail -5 gpurun_out/r03z_bench_$name.err; }; }
b default
b logged_200 --no-cpu-baseline --log --steps 200
b unlogged_200 --no-cpu-baseline --steps 200
b logged_50 --no-cpu-baseline --log
b unlogged_50 --no-cpu-baseline
b 2048 --no-cpu-baseline --envs-per-gpu 2048
b 1024 --no-cpu-baseline --envs-per-gpu 1024
b cfg2_8192_grid --no-cpu-baseline --envs-per-gpu 8192 --terrain grid --steps 30
b 16384_flat --no-cpu-baseline --envs-per-gpu 16384 --steps 20
b gloo2_same_device --no-cpu-baseline --gpus 2 --backend gloo --same-device --steps 20
b rccl_1rank --no-cpu-baseline --force-dist --steps 30
for f in gpurun_out/r03z_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f'.split('bench_')[1], round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| step', round(d['roofline']['launch_ms']*1e3,1), 'frac', round(d['roofline']['frac'],4), '| upd', round(d['roofline_update']['launch_ms']*1e3,1), round(d['roofline_update']['frac'],4), '| ar', c.get('grad_allreduce_us'), 'ckpt', c.get('end_of_learn_checkpoint_ms'))
" 2>&1 | tail -1; done
bash tools/prof_bench.sh r03 > gpurun_out/r03z_prof.log 2>&1; head -14 gpurun_out/kernel_stats_r03.csv | cut -c1-150
timeout 300 python tools/dist_overhead.py 2>&1 | grep "^(" | tee gpurun_out/r03z_dist_overhead.txt
