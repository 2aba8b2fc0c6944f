# round 6, final measurement pass: the whole GPU suite, the rocprofv3 statistics + counters of the bench loop, THEN the bench lines
# (so that the lines' counters_source is the counter summary of the very kernel they time)
R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -4
bash tools/r06_profile.sh 2>&1 | tail -34
cp gpurun_out/r06p_counters.json profiles/step_kernel_counters.json
bash tools/r06_final.sh 2>&1 | tail -20
python tools/time_hist_train.py 2>&1 | tail -1
