#!/bin/bash
# PMC passes for the step kernel (separate rocprofv3 runs, counters only; no trace domains mixed in).
# usage on the GPU box: bash tools/pmc_run.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- $CMD > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "env_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(tot):
        o.write(f"{k:28s} per-launch {tot[k]/cnt[k]:16.1f}   launches {cnt[k]}\n")
print(open("$OUT/summary.txt").read())
PY
