R=$GRAFT_REPO_ROOT
cd $R
L=$R/deep-whole-body-control_amd/wbc_amd
echo "== walk stamps (timing)"; WBC_XSTAMPS=walk WBC_STAMPS=1 python tools/time_step.py 1024 100 base 2>&1 | grep "walk stamps\|substep phase" | cut -c1-400
echo "== walk stamps (no stores)"; cp $L/libwbc_amd_timing.so /tmp/t.so; cp $L/libwbc_amd_wnostore.so $L/libwbc_amd_timing.so; WBC_XSTAMPS=walk WBC_STAMPS=1 python tools/time_step.py 1024 100 base 2>&1 | grep "walk stamps\|substep phase" | cut -c1-400; cp /tmp/t.so $L/libwbc_amd_timing.so
for N in 1024 4096; do for rep in 1 2; do
  echo "== product N=$N"; python tools/time_step.py $N 200 base 2>&1 | grep "step kernel" | cut -c1-200
  echo "== noatomic N=$N"; WBC_AMD_LIB=$L/libwbc_amd_noatomic.so python tools/time_step.py $N 200 base 2>&1 | grep "step kernel" | cut -c1-200
done; done
