R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py ppotiming -DWBC_PPO_TIMING 2>&1 | tail -1
python tools/time_act_layers.py 1024 2>&1 | grep "rows\|layer"; timeout 900 python -m pytest tests/test_gpu_policy_kernel.py -m gpu -x -q 2>&1 | tail -1; python tools/time_act.py 2>&1 | grep "rows\|median"
python tools/time_act_layers.py 4096 2>&1 | grep "rows\|layer"
