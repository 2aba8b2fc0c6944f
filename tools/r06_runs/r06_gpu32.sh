R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do for d in 0 1; do echo -n "WBC_NO_DEAL=$d "; WBC_NO_DEAL=$d python tools/time_step.py 1024 400 2>&1 | grep "step kernel"; done; done
for rep in 1 2 3; do for d in 0 1; do
WBC_NO_DEAL=$d python bench.py --envs-per-gpu 1024 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 1024 WBC_NO_DEAL=$d', round(d['value']), d['ms_per_step'], d['config']['collection_ms'], d['config']['learn_ms'], d['roofline']['launch_ms'])"
done; done
