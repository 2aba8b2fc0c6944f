R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py wavetiming -DWBC_WAVE_TIMING 2>&1 | tail -1
python tools/wave_bench_state.py 4096 2>&1 | grep "^busy_max\|^busy_mean\|^nh4\|^hit\|^miss\|^hshare\|unhinted"
