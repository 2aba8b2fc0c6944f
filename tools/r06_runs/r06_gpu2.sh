# round 6: parity of the sim kernels + step timing at three shard sizes (bash tools/r06_gpu2.sh)
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_sim_parity.py tests/test_gpu_contact_physics.py tests/test_wg_golden.py tests/test_gpu_env_runner.py -m gpu -x -q 2>&1 | tail -15
for N in 1024 2048 4096; do
  echo "== product N=$N"; python tools/time_step.py $N 200 base 2>&1 | grep "step kernel\|simulate kernel" | cut -c1-200
done
echo "== timing N=1024"; WBC_XSTAMPS=1 WBC_STAMPS=1 python tools/time_step.py 1024 100 base 2>&1 | grep -v "^$" | cut -c1-600
