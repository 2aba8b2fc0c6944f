R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py wavetiming -DWBC_WAVE_TIMING 2>&1 | tail -1
python tools/wave_bench_state.py 4096 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -110 | cut -c1-260
