cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sim_parity.py tests/test_wg_golden.py tests/test_gpu_terrain.py -m gpu -q -x --timeout 180 > gpurun_out/r03e_sim_tests.log 2>&1; rc=$?; tail -40 gpurun_out/r03e_sim_tests.log
if [ $rc -ne 0 ]; then echo "sim tests failed rc=$rc: stopping here"; exit 0; fi
timeout 900 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -15 > gpurun_out/r03e_all_tests.log; tail -8 gpurun_out/r03e_all_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench_default.json 2> gpurun_out/r03e_bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/r03e_bench_default.json').read().strip().splitlines()[0]); c=d['config']
print(round(d['value']), round(d['ms_per_step'],3), 'ms; collect', round(c['collection_ms'],2), 'learn', round(c['learn_ms'],2), 'step us', round(d['roofline']['launch_ms']*1e3,1), 'upd us', round(d['roofline_update']['launch_ms']*1e3,1))"
timeout 600 python tools/train_walk.py 2000 survive=2.0 energy=-6e-6 contacts_z=-1e-5 > gpurun_out/r03e_walk_K.jsonl 2> gpurun_out/r03e_walk_K.err; tail -1 gpurun_out/r03e_walk_K.jsonl | cut -c1-1000
