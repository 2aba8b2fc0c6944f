# round 6, first GPU call: the baseline's phase stamps at one / two / four waves per SIMD (bash tools/r06_gpu1.sh)
R=$GRAFT_REPO_ROOT
cd $R
for N in 1024 2048 4096; do
  echo "== product N=$N"; python tools/time_step.py $N 200 base 2>&1 | grep -v "^$" | cut -c1-400
  echo "== timing N=$N"; WBC_STAMPS=1 python tools/time_step.py $N 100 base 2>&1 | grep -v "^$" | cut -c1-600
done
