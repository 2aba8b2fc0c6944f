R=$GRAFT_REPO_ROOT
cd $R
timeout 280 python bench.py > gpurun_out/r02d_bench_default.json 2> gpurun_out/r02d_bench_default.err
timeout 280 python bench.py --no-cpu-baseline --envs-per-gpu 8192 --terrain grid > gpurun_out/r02d_bench_cfg2_8192_grid.json 2> gpurun_out/r02d_cfg2.err
timeout 280 python bench.py --no-cpu-baseline --envs-per-gpu 16384 --steps 20 > gpurun_out/r02d_bench_16384_flat.json 2> gpurun_out/r02d_16384.err
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --no-cpu-baseline --steps 20 > gpurun_out/r02d_bench_torchrun_1rank.json 2> gpurun_out/r02d_torchrun.err
for f in gpurun_out/r02d_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(' ', round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', d['config']['workload'][:60], d['config'].get('rccl_ranks'), d['config'].get('grad_allreduce_us'), 'roofline', round(d['roofline']['frac'],4), round(d.get('roofline_update',{}).get('frac',0),4))
" 2>&1 | tail -2; done
tail -3 gpurun_out/r02d_cfg2.err gpurun_out/r02d_torchrun.err | grep -v amdgpu | tail -8
bash tools/pmc_traffic.sh 2>&1 | tail -3
bash tools/pmc_traffic_ppo.sh 2>&1 | tail -8
