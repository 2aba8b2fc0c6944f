# XCD-aware env mapping of wbc_step_kernel against the identity mapping (tools/build_variant.py noxcd -DWBC_NO_XCD_MAP): time and HBM counters
cd $GRAFT_REPO_ROOT
bash tools/ab_flags.sh noxcd
for v in "" noxcd; do
  echo "== traffic ${v:-product}"
  WBC_AMD_LIB=$GRAFT_REPO_ROOT/deep-whole-body-control_amd/wbc_amd/libwbc_amd${v:+_$v}.so bash tools/pmc_traffic.sh
done
