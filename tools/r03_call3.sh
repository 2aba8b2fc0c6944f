cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python tools/train_walk.py 2000 "$@" > gpurun_out/r03c_walk_$name.jsonl 2> gpurun_out/r03c_walk_$name.err; echo "== $name $@"; tail -1 gpurun_out/r03c_walk_$name.jsonl | cut -c1-900; }
run A survive=1.0 lin_l1=2.0
run B survive=0.5 lin_l1=2.0 contacts_z=-2e-5
run C survive=1.0 lin_l1=4.0
run D survive=2.0 lin_l1=4.0
run E survive=0.2 lin_l1=2.0
