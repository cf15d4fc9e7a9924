#!/bin/bash
# Kernel trace + PMC passes for the step kernel of one bench workload (separate rocprofv3 runs; counters never mixed with trace
# domains).  usage on the GPU box:  bash tools/pmc_run.sh <tag> [workload] [steps]
# Writes gpurun_out/prof_<tag>_<workload>/{kernel_stats.csv,kernel_trace_summary.txt,pmc_summary.txt}; copy what is to be judged
# into profiles/.
set -u
TAG=${1:-r03}
WL=${2:-light_flat}
STEPS=${3:-40}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup 10 --no-cpu-baseline --no-rollout"
# 1. kernel trace + stats (per-kernel durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
grep '^{' $OUT/trace.log | tail -n 1 > $OUT/bench_line.json
# 2. counters, one pass per set
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $ARGS > $OUT/pass$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
# kernel stats: per kernel name calls, total, average (ns)
rows = collections.defaultdict(list)
for f in glob.glob(out + "/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/kernel_trace_summary.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 10 --no-cpu-baseline --no-rollout\n")
    o.write(f"{'kernel':110s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s}\n")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"{k[:110]:110s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e6:10.3f}\n")
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    os.replace(f, out + "/kernel_stats.csv")
def family(name):
    for f in ("env_narrow_kernel", "env_step_kernel", "env_fixup_kernel", "env_kernel"):
        if f in name:
            return f
    return None
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam:
            tot[fam][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[fam][r["Counter_Name"]] += 1
# the kernel with the most GPU time first (bench.py reads the first section as the dominant kernel)
dur = collections.defaultdict(float)
for k, v in rows.items():
    if family(k):
        dur[family(k)] += sum(v)
with open(out + "/pmc_summary.txt", "w") as o:
    for fam in sorted(tot, key=lambda f: -dur[f]):
        o.write(f"per launch of cosim::{fam}, rocprofv3 --pmc passes over: bench.py --workload $WL --steps $STEPS --warmup 10\n")
        for k in sorted(tot[fam]):
            o.write(f"{k:28s} per-launch {tot[fam][k]/cnt[fam][k]:16.1f}   launches {cnt[fam][k]}\n")
print(open(out + "/kernel_trace_summary.txt").read()[:3000])
print(open(out + "/pmc_summary.txt").read())
print(open(out + "/bench_line.json").read()[:600])
PY
