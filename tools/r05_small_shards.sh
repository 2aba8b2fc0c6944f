# kernel statistics of the bench loop at small shards (what every GPU of a strong-scaled 16384-env job gets): bash tools/r05_small_shards.sh
R=$GRAFT_REPO_ROOT
for E in 1024 2048; do
  bash $R/tools/r05_stats.sh --envs-per-gpu $E --steps 30 > $R/gpurun_out/small_$E.txt 2>&1
  cp $R/gpurun_out/r05p_bench_kernel_stats.csv $R/gpurun_out/r05_bench_kernel_stats_$E.csv
  echo "== $E"; cat $R/gpurun_out/small_$E.txt | cut -c1-110
done
