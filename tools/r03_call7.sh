cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/train_walk.py 7000 survive=2.0 energy=-6e-6 contacts_z=-1e-5 sched=scratch out=gpurun_out/r03g_walk_scratch7000.pt > gpurun_out/r03g_walk_scratch7000.jsonl 2> gpurun_out/r03g_walk.err; tail -2 gpurun_out/r03g_walk_scratch7000.jsonl | cut -c1-1200
timeout 600 python -m pytest tests/test_gpu_learning.py -m gpu -q -x -s --timeout 300 2>&1 | grep -v "amdgpu.ids" | tail -12
