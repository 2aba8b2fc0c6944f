# round 5, GPU call 1: the -m gpu suite on the 4-sweep solver, then the bench line with the solver's sweeps at 4 (reference) and 2
# (round 4), in the default and the standing regime
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu1_tests.log 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r05_gpu1_tests.log
python bench.py --steps 50 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --steps 50 --contact-iters 2 --no-cpu-baseline > gpurun_out/r05_bench_iters2.json 2>/dev/null
python bench.py --steps 50 --regime standing --no-cpu-baseline > gpurun_out/r05_bench_standing.json 2>/dev/null
python bench.py --steps 50 --regime standing --contact-iters 2 --no-cpu-baseline > gpurun_out/r05_bench_standing_iters2.json 2>/dev/null
for f in default iters2 standing standing_iters2; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r05_bench_$f.json"))
    print("$f", round(d["value"]/1e6,3), "M  ms", round(d["ms_per_step"],3), "coll", round(d["config"]["collection_ms"],3), "learn", round(d["config"]["learn_ms"],3), "step_us", round(d["roofline"]["launch_ms"]*1e3,1), "upd", round(d["roofline_update"]["launch_ms"]*1e3,1))
except Exception as e: print("$f", "ERR", e)
PY
done
