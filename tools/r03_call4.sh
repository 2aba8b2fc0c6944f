cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python tools/train_walk.py 2000 "$@" > gpurun_out/r03d_walk_$name.jsonl 2> gpurun_out/r03d_walk_$name.err; echo "== $name $@"; tail -1 gpurun_out/r03d_walk_$name.jsonl | cut -c1-1100; tail -2 gpurun_out/r03d_walk_$name.err | grep -v amdgpu; }
run F lin_l1=0 lin_exp=1.5 survive=1.0 energy=-6e-6 contacts_z=-1e-5
run G lin_l1=0 lin_exp=1.5 survive=0.2 energy=-6e-6 contacts_z=-1e-5
run H lin_l1=0 lin_exp=1.5 survive=1.0
run I lin_l1=0 lin_exp=3.0 survive=1.0 energy=-6e-6 contacts_z=-1e-5
run J lin_l1=0 lin_exp=1.5 survive=1.0 energy=0 contacts_z=0
run K survive=2.0 energy=-6e-6 contacts_z=-1e-5
