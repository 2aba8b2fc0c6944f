# rocprofv3 kernel statistics of the bench command (40 iterations: 2 DAgger iterations inside) -> gpurun_out/prof_<tag>/, summary CSV
# usage (on the GPU box, from the repo root): bash tools/prof_bench.sh <tag> [bench args...]
TAG=${1:-r02}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-cpu-baseline --steps 40 "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
f=$(ls $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | tail -1)
cp "$f" $R/gpurun_out/kernel_stats_$TAG.csv
head -25 $R/gpurun_out/kernel_stats_$TAG.csv | cut -c1-200
