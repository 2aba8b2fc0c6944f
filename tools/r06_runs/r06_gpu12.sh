R=$GRAFT_REPO_ROOT
cd $R
for V in dec0 dec2 dec8; do echo "== insts $V"; bash tools/r06_pmc_insts.sh 4096 $V 2>&1 | grep wbc_step; done
