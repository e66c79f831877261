#!/bin/bash
# SQ counters of the normal-field kernels at N = 2048: the closed form (product) against round 4's literal evaluation
# (variants/normals_literal.so) -- instruction counts and VALU-active cycles per dispatch.   tools/gpu_pmc_normals.sh <tag>
set -u
exec < /dev/null
TAG=${1:-pmcn}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for lib in libocean_hip.so variants/normals_literal.so; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
    i=$((i+1)); D=$O/$(basename $lib .so)_p$i
    OCEAN_HIP_LIB=$GRAFT_REPO_ROOT/gfx_ocean_amd/$lib timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o run -- python $GRAFT_REPO_ROOT/tools/normals_time.py 2048 > $D.stdout.txt 2>&1
    python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ocean::", "")[:28]
        a = acc[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if "normals" in k: print(f"$(basename $lib .so)".ljust(18), f"{k:28s} {c:22s} {v / n:16.1f}  ({n} dispatches)")
PY
  done
done | tee $O/sq_counters_normals.txt
find $O -name "*.csv" -size +1M -delete
