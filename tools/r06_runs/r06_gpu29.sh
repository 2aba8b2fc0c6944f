R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_wg_golden.py tests/test_gpu_sim_parity.py tests/test_gpu_env_runner.py tests/test_gpu_deal.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -3 | cut -c1-400
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), d['config']['collection_ms'], d['roofline']['launch_ms'])"
