# learner parity tests, then the bench loop at three shard sizes (GPU box, repo root)
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_ppo_kernel.py tests/test_ppo_parity.py tests/test_gpu_env_runner.py tests/test_kernel_codegen.py -m gpu -q -x 2>&1 | tail -4
for E in 4096 2048 1024; do
  timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $E 2>/dev/null > gpurun_out/r05_bench_$E.json
  python - <<PY
import json; d=json.loads(open("gpurun_out/r05_bench_$E.json").read().strip().splitlines()[-1])
print('envs $E', round(d['value']/1e6,3), 'M  iter', round(d['ms_per_step'],3), 'collect', round(d['config']['collection_ms'],3), 'learn', round(d['config']['learn_ms'],3), 'mb call', round(d['roofline_update']['launch_ms']*1000,1), 'frac', round(d['roofline_update']['frac'],3))
PY
done
