# round 4, first GPU call: the new step kernel (box actor, mid-shanks, 47 contacts, friction fix) against the oracle + timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -rs --deselect tests/test_gpu_learning.py > gpurun_out/r04a_gputest.log 2>&1
grep -a "passed\|failed\|SKIPPED\|Error\|error" gpurun_out/r04a_gputest.log | head -20
tail -40 gpurun_out/r04a_gputest.log
for n in 1024 4096 16384; do WBC_STAMPS=1 timeout 200 python tools/time_step.py $n 200 base 2>&1 | grep -v "^$" | tail -8; done > gpurun_out/r04a_time_step.txt 2>&1
cat gpurun_out/r04a_time_step.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err || tail -5 gpurun_out/r04a_bench.err
cat gpurun_out/r04a_bench.json | cut -c1-600
