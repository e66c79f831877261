#!/bin/bash
# LDS-DMA loader, second measurement: A/B incl. the inter-workgroup-duplicate ablation and the loader at 2048, then
# rocprofv3 stats + HBM counters of the shipped build.   tools/gpu_dma2.sh <tag>
set -u
exec < /dev/null
TAG=${1:-dma2}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== A/B"; bash tools/ab_variants.sh $TAG 2048 4096 8192 2>&1 | tail -60
cd /tmp
run_prof() {   # name, N, traffic flag ("-", f16 or staged), then the command
  local name=$1 n=$2 flag=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/stats -o run -- "$@" > $O/$name.stats_stdout.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$name/pmc_$c -o run -- "$@" > $O/$name.pmc_${c}_stdout.txt 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/$name > $O/$name.summary.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/make_hbm_traffic.py $O/$name $n $TAG $flag > $O/$name.hbm_traffic.txt 2>&1
  cp $GRAFT_REPO_ROOT/profiles/hbm_traffic_*.json $O/ 2>/dev/null
  echo "== $name"; grep -v "^$" $O/$name.summary.txt | cut -c1-160 | head -12; cat $O/$name.hbm_traffic.txt
}
run_prof fused_n4096 4096 - python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 5 --profile-frames 5
run_prof fused_n8192 8192 - python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 8192 --steps 60 --warmup 2 --profile-frames 2
run_prof fused_n8192_f16 8192 f16 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 8192 --spectrum f16 --steps 60 --warmup 2 --profile-frames 2
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" -size +2M -delete
