R=$GRAFT_REPO_ROOT
cd $R
for V in base nocontact dec1 noreset; do echo "== insts $V"; bash tools/r06_pmc_insts.sh 4096 $V; done
