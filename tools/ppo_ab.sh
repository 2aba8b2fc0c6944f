# A/B of the PPO minibatch kernels on the GPU box: rocprofv3 kernel statistics of tools/time_ppo.py for each library variant / switch
# usage: bash tools/ppo_ab.sh "<label>:<env assignments>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  env $envs WBC_ITERS=30 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ppo_ab_$label -- python $R/tools/time_ppo.py 40960 > $R/gpurun_out/ppo_ab_$label.log 2>&1
  f=$(ls $R/gpurun_out/ppo_ab_$label/*/*kernel_stats.csv | tail -1)
  echo "$label: $(grep -E 'ppo_chain|ppo_fwd_bwd16|ppo_wgrad|chain_pack|pack16' $f | awk -F, '{gsub(/"/,""); printf "%s avg_us=%.1f  ", substr($1,1,24), $4/1000}')"
done
