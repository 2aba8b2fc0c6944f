# round 3: the measurements DESIGN.md / profiles/ quote. usage (GPU box, repo root): bash tools/r03_final.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rs 2>&1 | tail -6 > gpurun_out/r03z_gputest.log; cat gpurun_out/r03z_gputest.log
b() { name=$1; shift; timeout 400 python bench.py "$@" > gpurun_out/r03z_bench_$name.json 2> gpurun_out/r03z_bench_$name.err || echo "bench $name FAILED rc=$?"; }
b default
b unlogged_a --no-cpu-baseline
b logged_a --no-cpu-baseline --log
b unlogged_b --no-cpu-baseline
b logged_b --no-cpu-baseline --log
b 2048 --no-cpu-baseline --envs-per-gpu 2048
b 1024 --no-cpu-baseline --envs-per-gpu 1024
b cfg2_8192_grid --no-cpu-baseline --envs-per-gpu 8192 --terrain grid --steps 30
b 16384_flat --no-cpu-baseline --envs-per-gpu 16384 --steps 20
b gloo2_same_device --no-cpu-baseline --gpus 2 --backend gloo --same-device --steps 20
b rccl_1rank --no-cpu-baseline --force-dist --steps 30
for f in gpurun_out/r03z_bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[0]); c=d['config']
print('$f'.split('bench_')[1], round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), '| step', round(d['roofline']['launch_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],4), '| update', round(d.get('roofline_update',{}).get('launch_ms',0)*1e3,1), 'us frac', round(d.get('roofline_update',{}).get('frac',0),4), '| allreduce us', c.get('grad_allreduce_us'))
" 2>&1 | tail -1; done
bash tools/prof_bench.sh r03 > gpurun_out/r03z_prof.log 2>&1; head -16 gpurun_out/kernel_stats_r03.csv | cut -c1-160
bash tools/pmc_step.sh > gpurun_out/r03z_pmc_step_sq.txt 2>&1; cat gpurun_out/r03z_pmc_step_sq.txt | tail -20
bash tools/pmc_traffic.sh > gpurun_out/r03z_pmc_traffic.txt 2>&1; cat gpurun_out/r03z_pmc_traffic.txt | tail -3
WBC_STAMPS=1 timeout 300 python tools/time_step.py 4096 200 base 2>&1 | grep -v amdgpu.ids > gpurun_out/r03z_time_step.txt
timeout 200 python tools/time_step.py 1024 300 base 2>&1 | grep "^base" >> gpurun_out/r03z_time_step.txt
timeout 200 python tools/time_step.py 16384 100 base 2>&1 | grep "^base" >> gpurun_out/r03z_time_step.txt
cat gpurun_out/r03z_time_step.txt
timeout 300 python tools/dist_overhead.py 2>&1 | grep "^(" | tee gpurun_out/r03z_dist_overhead.txt
