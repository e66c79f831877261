#!/bin/bash
# real-output pass 2 as shipped (N >= 8192): GPU tests that touch it, bench lines, timelines.   tools/gpu_real2b.sh <tag>
set -u
exec < /dev/null
TAG=${1:-real2b}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_gpu_race.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.txt
for cfg in "8192:" "8192:--spectrum f16" "8192:--spectrum f16 --intermediate bfp16" "16384:"; do
  n=${cfg%%:*}; fl=${cfg#*:}
  timeout 600 python bench.py --no-cpu-baseline --n $n $fl --steps 60 --warmup 3 2> /dev/null | tail -1 > $O/bench_n${n}_$(echo $fl | tr -d ' -').json
  python - <<PY
import json
d=json.loads(open("$O/bench_n${n}_$(echo $fl | tr -d ' -').json").read())
print("$n $fl", round(d['value'],1), round(d['ms_per_step'],4), [(k['name'], round(k['avg_ms']*1000,1), round(k['frac'],3)) for k in d['roofline']['kernels']])
PY
done
for n in 8192 16384; do timeout 300 ./tools/timeline $n > $O/timeline_n$n.txt 2>&1; grep -A 12 "^pass 2" $O/timeline_n$n.txt | cut -c1-150; done
