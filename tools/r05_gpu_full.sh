R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r05_gpu_full_tests.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error" gpurun_out/r05_gpu_full_tests.log | tail -3
python bench.py --steps 50 > gpurun_out/r05_bench_default.json 2>/dev/null
python bench.py --steps 50 --regime standing --no-cpu-baseline > gpurun_out/r05_bench_standing.json 2>/dev/null
for f in default standing; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_bench_$f.json"))
    print("$f", round(d["value"]/1e6,3), "M  ms", round(d["ms_per_step"],3), "coll", round(d["config"]["collection_ms"],3), "learn", round(d["config"]["learn_ms"],3), "step_us", round(d["roofline"]["launch_ms"]*1e3,1), "upd", round(d["roofline_update"]["launch_ms"]*1e3,1))
except Exception as e: print("$f", "ERR", e)
PY
done
