R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_sim_parity.py tests/test_gpu_deal.py tests/test_gpu_env_runner.py tests/test_gpu_contact_physics.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -3 | cut -c1-400
for n in 1024 2048 4096; do python tools/time_step.py $n 200 2>&1 | grep "step kernel"; done
WBC_ACT_SCALE=1.0 python tools/time_step.py 4096 200 2>&1 | grep "step kernel"
WBC_ACT_SCALE=1.0 python tools/time_step.py 1024 200 2>&1 | grep "step kernel"
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
done
python bench.py --envs-per-gpu 1024 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench1024', d['value'], d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
timeout 600 python -m pytest tests/test_gpu_contact_physics.py -m gpu -q -s -k sticks 2>&1 | grep "sticks:\|passed\|failed"
