# A/B: the step kernel of this tree against a snapshot of an earlier commit in scratch_r03/ (not committed; make it with
#   mkdir scratch_r03 && git archive <commit> | tar -x -C scratch_r03 && (cd scratch_r03 && python -c 'import __graft_entry__ as g; g.build()')
# ), same scenarios. For variants of THIS tree use tools/build_variant.py + tools/ab_flags.sh.
cd $GRAFT_REPO_ROOT
for sc in "0.5 20" "0.0 120"; do set -- $sc
  for d in scratch_r03 .; do
    ( cd $d; echo "== $d act_scale=$1 warm=$2"; WBC_ACT_SCALE=$1 WBC_WARM_STEPS=$2 timeout 200 python tools/time_step.py 4096 200 base 2>&1 | grep "step kernel"; WBC_ACT_SCALE=$1 WBC_WARM_STEPS=$2 timeout 200 python tools/time_step.py 16384 100 base 2>&1 | grep "step kernel" )
  done
done
