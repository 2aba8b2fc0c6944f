R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_policy_kernel.py tests/test_gpu_env_runner.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -3 | cut -c1-400
for m in 0 1073741824; do echo "== WBC_ACT16_SPLIT_MAX_ROWS=$m"; WBC_ACT16_SPLIT_MAX_ROWS=$m python tools/time_act.py 2>&1 | grep "rows\|median"; done
for m in 0 2048; do for n in 1024 2048; do
WBC_ACT16_SPLIT_MAX_ROWS=$m python bench.py --envs-per-gpu $n --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench split<=$m envs $n', round(d['value']), d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'])"
done; done
