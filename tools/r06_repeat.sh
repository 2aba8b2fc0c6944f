# run-to-run spread of the headline on ONE box: three default runs (50 / 5) and three driver-style runs (20 / 5), then the same
# driver-style run three times with the per-launch deal switched off (WBC_NO_DEAL=1). GPU box, repo root.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r06_bench_repeat.txt
run() {
  timeout 300 python bench.py --no-cpu-baseline $2 2>/dev/null < /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$1 $2:', round(d['value']/1e6,3), 'M env-steps/s,', round(d['ms_per_step'],3), 'ms per iteration (collect', round(c['collection_ms'],2), '+ learn', str(round(c['learn_ms'],2))+'), step kernel', round(d['roofline']['launch_ms']*1e3,1), 'us, minibatch call', round(d['roofline_update']['launch_ms']*1e3,1), 'us')" | tee -a gpurun_out/r06_bench_repeat.txt
}
for args in "--steps 50 --warmup 5" "--steps 20 --warmup 5" "--steps 50 --warmup 5" "--steps 20 --warmup 5" "--steps 50 --warmup 5" "--steps 20 --warmup 5"; do run "deal on " "$args"; done
for i in 1 2 3; do WBC_NO_DEAL=1 run "deal off" "--steps 20 --warmup 5"; done
