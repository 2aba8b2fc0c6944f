cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 8192 16384 24576 32768 40960 49152; do
  WBC_ITERS=30 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ppo_sz_$B -- python $R/tools/time_ppo.py $B > /dev/null 2>&1
  f=$(ls $R/gpurun_out/ppo_sz_$B/*/*kernel_stats.csv | tail -1)
  echo "B=$B units=$((B/8)) $(grep -E 'ppo_chain|ppo_wgrad' $f | awk -F, '{gsub(/"/,""); printf "%s avg_us=%.1f  ", $1, $4/1000}')"
done
