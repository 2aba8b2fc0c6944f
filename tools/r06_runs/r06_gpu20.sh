R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py wavetiming -DWBC_WAVE_TIMING 2>&1 | tail -1
echo "== dealt"; python tools/wave_map.py 4096 2>&1 | grep "^prev\|^launch\|SIMDs by\|fraction"
echo "== not dealt"; WBC_NO_DEAL=1 python tools/wave_map.py 4096 2>&1 | grep "^launch\|SIMDs by"
