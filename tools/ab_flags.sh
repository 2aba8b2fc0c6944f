# A/B of compiler-flag variants of the library (tools/build_variant.py <name> <flags>): step kernel alone, same scenario
cd $GRAFT_REPO_ROOT
W=deep-whole-body-control_amd/wbc_amd
for v in "" $@; do
  lib=$W/libwbc_amd${v:+_$v}.so
  for rep in 1 2; do
    echo "== ${v:-product} rep $rep"; WBC_AMD_LIB=$PWD/$lib timeout 200 python tools/time_step.py 4096 200 base 2>&1 | grep "step kernel"
  done
done
