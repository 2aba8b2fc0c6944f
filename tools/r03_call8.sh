cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- default"; WBC_ONLY_C=1 timeout 300 python tools/dist_overhead.py 2>&1 | grep "^("
echo "--- TORCH_NCCL_AVOID_RECORD_STREAMS=1"; TORCH_NCCL_AVOID_RECORD_STREAMS=1 WBC_ONLY_C=1 timeout 300 python tools/dist_overhead.py 2>&1 | grep "^("
echo "--- no async stats"; WBC_NO_ASYNC_STATS=1 WBC_ONLY_C=1 timeout 300 python tools/dist_overhead.py 2>&1 | grep "^("
echo "--- GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 WBC_ONLY_C=1 timeout 300 python tools/dist_overhead.py 2>&1 | grep "^("
echo "--- NCCL_LAUNCH_MODE/blocking wait"; TORCH_NCCL_BLOCKING_WAIT=1 WBC_ONLY_C=1 timeout 300 python tools/dist_overhead.py 2>&1 | grep "^("
