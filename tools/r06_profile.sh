# round 6: the profiles bench.py's roofline block quotes, all collected ON THE BENCH LOOP (python bench.py ...), in separate passes as
# MI355X_MICROARCH.md prescribes (kernel statistics; FETCH_SIZE; WRITE_SIZE; the SQ instruction counters). GPU box, repo root:
#   bash tools/r06_profile.sh      -> gpurun_out/r06p_*  (copy the summaries to profiles/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
BENCH="python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06p_stats -- python $R/bench.py --no-cpu-baseline --steps 40 > $R/gpurun_out/r06p_stats_bench.json 2> $R/gpurun_out/r06p_stats.err
cp $(ls $R/gpurun_out/r06p_stats/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r06p_bench_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r06p_pmc/$tag -- $BENCH > /dev/null 2> $R/gpurun_out/r06p_pmc_$tag.err
done
python - <<'PY'
import csv, glob, os, json, statistics, collections, subprocess, datetime
R = os.environ['GRAFT_REPO_ROOT']
acc = collections.defaultdict(list)
for f in sorted(glob.glob(R + '/gpurun_out/r06p_pmc/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for name in ('wbc_step_kernel', 'ppo_chain_kernel', 'ppo_wgrad_kernel', 'wbc_policy_act16_kernel'):
            if k.startswith(name):
                acc[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
def med(name, c):
    v = acc.get((name, c)); return statistics.median(v) if v else None
import sys
sys.path.insert(0, R)
from bench import step_kernel_sha16
out = {"kernel_sha16": step_kernel_sha16(), "contact_iters": 4, "collected_with": "tools/r06_profile.sh: rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --no-cpu-baseline --steps 10 --warmup 2",
       "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"), "num_envs": 4096, "kernels": {}}
for name in ('wbc_step_kernel', 'ppo_chain_kernel', 'ppo_wgrad_kernel', 'wbc_policy_act16_kernel'):
    d = {c: med(name, c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU",
                                   "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES")}
    d["launches_sampled"] = len(acc.get((name, "SQ_INSTS_VALU"), []))
    out["kernels"][name] = d
s = out["kernels"]["wbc_step_kernel"]
if s["FETCH_SIZE"] is not None:
    out["step_kernel"] = {
        "FETCH_SIZE_KB_per_launch_raw": s["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch_raw": s["WRITE_SIZE"],
        "hbm_bytes_per_launch_raw": 1024.0 * (s["FETCH_SIZE"] + s["WRITE_SIZE"]),
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports by 2x on gfx950 (calibrated on 16-B-per-lane streams)
        "hbm_bytes_per_launch": 1024.0 * (2.0 * s["FETCH_SIZE"] + s["WRITE_SIZE"]),
        "valu_insts_per_wave": s["SQ_INSTS_VALU"] / s["SQ_WAVES"], "salu_insts_per_wave": s["SQ_INSTS_SALU"] / s["SQ_WAVES"],
        "lds_insts_per_wave": s["SQ_INSTS_LDS"] / s["SQ_WAVES"], "active_lanes": s["SQ_THREAD_CYCLES_VALU"] / s["SQ_INSTS_VALU"],
        "wait_any_frac": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], "lds_bank_conflict_frac": s["SQ_LDS_BANK_CONFLICT"] / s["SQ_ACTIVE_INST_LDS"]}
json.dump(out, open(R + '/gpurun_out/r06p_counters.json', 'w'), indent=1)
print(json.dumps(out.get("step_kernel"), indent=1))
PY
head -16 $R/gpurun_out/r06p_bench_kernel_stats.csv | cut -c1-160
