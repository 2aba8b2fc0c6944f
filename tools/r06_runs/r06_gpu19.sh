R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ppo_kernel.py tests/test_ppo_parity.py -m gpu -x -q -s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep "passed\|failed\|Error\|post-update\|assert" | tail -12 | cut -c1-300
python tools/time_hist_train.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -2 | cut -c1-300
WBC_STAMPS=1 python tools/time_hist_train.py 2>&1 | tail -1 | cut -c1-600
