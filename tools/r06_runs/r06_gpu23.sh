R=$GRAFT_REPO_ROOT
cd $R
python tools/build_variant.py wavetiming -DWBC_WAVE_TIMING 2>&1 | tail -1
echo "== dealt"; python tools/wave_bench_state.py 4096 2>&1 | grep "^busy\|^nh4\|^hit\|^miss\|^hshare\|SIMD busy\|busiest"
echo "== not dealt"; WBC_NO_DEAL=1 python tools/wave_bench_state.py 4096 2>&1 | grep "^busy\|^nh4"
timeout 900 python -m pytest tests/test_gpu_deal.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -2 | cut -c1-400
for n in 1024 2048 4096; do python tools/time_step.py $n 200 2>&1 | grep "step kernel"; done
for i in 1 2; do
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
WBC_NO_DEAL=1 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench nodeal', d['value'], d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
done
