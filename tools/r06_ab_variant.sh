# same-box A/B of the product library against ONE build variant (-D flags as arguments): deal test on the variant, stand-alone step
# kernel, the bench's event pairs (three runs each).  usage: bash tools/r06_ab_variant.sh -DWBC_SOMETHING ...
R=$GRAFT_REPO_ROOT
cd $R
W=$R/deep-whole-body-control_amd/wbc_amd
python tools/build_variant.py abv "$@" 2>&1 | tail -1
WBC_AMD_LIB=$W/libwbc_amd_abv.so timeout 900 python -m pytest tests/test_gpu_deal.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do for v in "" _abv; do for n in 2048 4096; do echo -n "base$v "; WBC_AMD_LIB=$W/libwbc_amd$v.so python tools/time_step.py $n 300 2>&1 | grep "step kernel"; done; done; done
for rep in 1 2 3; do for v in "" _abv; do
WBC_AMD_LIB=$W/libwbc_amd$v.so python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench base$v', round(d['value']), d['config']['collection_ms'], d['roofline']['launch_ms'])"
done; done
