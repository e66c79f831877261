#!/bin/bash
# SQ-level counters of the two fused kernels at N = 4096 (LDS bank conflicts, instruction mix, busy cycles).  tools/gpu_pmc_sq.sh <tag>
set -u
exec < /dev/null
TAG=${1:-pmcsq}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 5 --profile-frames 2 --ramp-frames 20 --distribution-frames 20 > $O/p${i}_stdout.txt 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/p$i/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ocean::", "")[:40]
        a = acc[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if k.startswith("k_half"): print(f"{k:42s} {c:28s} {v / n:16.1f}  ({n} dispatches)")
PY
done | tee $O/sq_counters.txt
find $O -name "*.csv" -size +1M -delete
