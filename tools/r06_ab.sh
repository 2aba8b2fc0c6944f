# same-box A/B of two prebuilt libraries (libwbc_amd_old.so / libwbc_amd_new.so): deal test on the new one, stand-alone step kernel,
# the bench's event pairs
R=$GRAFT_REPO_ROOT
cd $R
W=$R/deep-whole-body-control_amd/wbc_amd
WBC_AMD_LIB=$W/libwbc_amd_new.so timeout 900 python -m pytest tests/test_gpu_deal.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do for v in old new; do for n in 1024 2048 4096; do echo -n "$v "; WBC_AMD_LIB=$W/libwbc_amd_$v.so python tools/time_step.py $n 300 2>&1 | grep "step kernel"; done; done; done
for rep in 1 2 3; do for v in old new; do
WBC_AMD_LIB=$W/libwbc_amd_$v.so python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', round(d['value']), d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
done; done
