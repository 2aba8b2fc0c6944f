# rocprofv3 kernel statistics of tools/time_ppo.py 40960 for library variants: bash tools/ab_ppo_prof.sh <variant> ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" $@ "" $@; do
  d=$R/gpurun_out/abppo_${v:-product}_$RANDOM
  WBC_AMD_LIB=$R/deep-whole-body-control_amd/wbc_amd/libwbc_amd${v:+_$v}.so WBC_ITERS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/tools/time_ppo.py 40960 > /dev/null 2>&1
  f=$(ls $d/*/*kernel_stats.csv | tail -1)
  echo "${v:-product}: $(grep -E 'ppo_chain|ppo_wgrad' $f | awk -F, '{gsub(/"/,""); printf "%s avg_us=%.1f  ", $1, $4/1000}')"
done
