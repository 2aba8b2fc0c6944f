# the whole GPU suite + the default bench line (bash tools/r06_gpu_full.sh)
R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -8
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 3000 gpurun_out/r06_bench_default.json
