cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sim_parity.py tests/test_wg_golden.py -m gpu -q -x --timeout 180 > gpurun_out/r03f_sim_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r03f_sim_tests.log
if [ $rc -ne 0 ]; then echo "sim tests failed rc=$rc: stopping here"; tail -60 gpurun_out/r03f_sim_tests.log; exit 0; fi
WBC_STAMPS=1 timeout 300 python tools/time_step.py 4096 200 base it1 nocontact 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03f_time_step.txt
timeout 300 python tools/time_step.py 4096 300 base 2>&1 | grep "^base" | tee -a gpurun_out/r03f_time_step.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03f_bench_default.json 2> gpurun_out/r03f_bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/r03f_bench_default.json').read().strip().splitlines()[0]); c=d['config']
print(round(d['value']), round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), 'step us', round(d['roofline']['launch_ms']*1e3,1), 'upd us', round(d['roofline_update']['launch_ms']*1e3,1))"
