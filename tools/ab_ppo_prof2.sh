# per-kernel A/B of library variants on the fused PPO minibatch call (rocprofv3 kernel stats over tools/time_ppo.py):
#   bash tools/ab_ppo_prof2.sh <B> <variant> ...     ("" = product)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B=$1; shift
W=$R/deep-whole-body-control_amd/wbc_amd
for v in product "$@" product "$@"; do
  lib=$W/libwbc_amd_$v.so; [ $v = product ] && lib=$W/libwbc_amd.so
  [ -f $lib ] || { echo "B=$B $v: $lib missing"; continue; }
  rm -rf $R/gpurun_out/abp_$v
  WBC_AMD_LIB=$lib WBC_ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abp_$v -- python $R/tools/time_ppo.py $B > $R/gpurun_out/abp_$v.log 2>&1 < /dev/null
  f=$(ls $R/gpurun_out/abp_$v/*/*kernel_stats.csv 2>/dev/null | tail -1)
  [ -n "$f" ] || { echo "B=$B $v: no kernel statistics (see gpurun_out/abp_$v.log)"; continue; }
  echo "B=$B $v: $(grep -E 'ppo_chain|ppo_wgrad|ppo_grad_reduce|ppo_reduce' $f | awk -F, '{gsub(/"/,""); printf "%s %.1f  ", $1, $4/1000}') | $(grep 'us per call' $R/gpurun_out/abp_$v.log | sed 's/.*: //')"
  rm -rf $R/gpurun_out/abp_$v
done
