# XCD-aware env mapping of wbc_step_kernel against the identity mapping: time and HBM counters (round 4; the WBC_NO_XCD_MAP switch it built with was removed in round 6, when the per-launch deal inside each XCD range went in -- kept for the record)
cd $GRAFT_REPO_ROOT
bash tools/ab_flags.sh noxcd
for v in "" noxcd; do
  echo "== traffic ${v:-product}"
  WBC_AMD_LIB=$GRAFT_REPO_ROOT/deep-whole-body-control_amd/wbc_amd/libwbc_amd${v:+_$v}.so bash tools/pmc_traffic.sh
done
