R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_deal.py -m gpu -x -q 2>&1 | tail -1
python bench.py --envs-per-gpu 1024 > gpurun_out/r06z_bench_1024.json 2> /dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r06z_bench_1024.json').read().strip().splitlines()[-1]); print('bench 1024', round(d['value']), d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
