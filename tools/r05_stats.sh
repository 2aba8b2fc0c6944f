# kernel statistics of the bench loop (rocprofv3 --kernel-trace --stats), GPU box, repo root: bash tools/r05_stats.sh [bench args]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/r05p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05p_stats -- python $R/bench.py --no-cpu-baseline --steps 40 "$@" > $R/gpurun_out/r05p_stats_bench.json 2> $R/gpurun_out/r05p_stats.err
cp $(ls $R/gpurun_out/r05p_stats/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05p_bench_kernel_stats.csv
head -14 $R/gpurun_out/r05p_bench_kernel_stats.csv | cut -c1-150
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r05p_stats_bench.json")); print(round(d["value"]/1e6,3), d["ms_per_step"], d["config"]["collection_ms"], d["config"]["learn_ms"])
PY
