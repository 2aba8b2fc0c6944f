R=$GRAFT_REPO_ROOT
cd $R
for N in 1024 4096; do echo "=== wave spread N=$N"; python tools/wave_spread.py $N 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -22 | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_sim_parity.py -m gpu -x -q -k "box_actor" -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep "box actor\|passed\|failed\|Error\|assert" | cut -c1-400 | tail -20
timeout 900 python -m pytest tests/test_ppo_parity.py -m gpu -x -q -s 2>&1 | grep "post-update\|passed\|failed" | cut -c1-300
