R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py wt_a -DWBC_WAVE_TIMING -DWBC_SOON_A=0.02f -DWBC_SOON_B=6.f 2>&1 | tail -1
for v in wt_a; do echo "== $v"; WBC_WAVE_LIB=$v python tools/wave_bench_state.py 4096 2>&1 | tail -25; done
