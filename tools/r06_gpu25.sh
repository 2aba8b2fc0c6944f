R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py prio -DWBC_DEAL_PRIO 2>&1 | tail -1
for i in 1 2 3; do
for v in "" _prio; do
WBC_AMD_LIB=$R/deep-whole-body-control_amd/wbc_amd/libwbc_amd$v.so python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench$v', d['value'], d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
done; done
for v in "" _prio; do for n in 2048 4096; do WBC_AMD_LIB=$R/deep-whole-body-control_amd/wbc_amd/libwbc_amd$v.so python tools/time_step.py $n 200 2>&1 | grep "step kernel"; done; done
