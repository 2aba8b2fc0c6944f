R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py ppotiming -DWBC_PPO_TIMING 2>&1 | tail -1
python tools/time_act_layers.py 1024 2>&1 | grep "rows\|layer"
python tools/time_act_layers.py 4096 2>&1 | grep "rows\|layer"
