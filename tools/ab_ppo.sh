# A/B of library variants on the fused PPO minibatch step (tools/time_ppo.py, B = 40960): bash tools/ab_ppo.sh <variant> ...
cd $GRAFT_REPO_ROOT
W=deep-whole-body-control_amd/wbc_amd
for v in "" $@ "" $@; do
  echo "== ${v:-product}"; WBC_AMD_LIB=$PWD/$W/libwbc_amd${v:+_$v}.so timeout 200 python tools/time_ppo.py 40960 2>&1 | tail -n 3
done
