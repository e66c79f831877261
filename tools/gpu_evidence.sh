#!/bin/bash
# Evidence run of a round for profiles/: bench lines (headline, config 5, config 3 with and without the normal field, config 2 plain
# and batched, 8192, 16384), rocprofv3 kernel stats + HBM counters (FETCH_SIZE / WRITE_SIZE in separate passes) of the same commands.
#   tools/gpu_evidence.sh <tag>
set -u
exec < /dev/null
TAG=${1:-ev}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python $GRAFT_REPO_ROOT/bench.py"
echo "== bench (driver flags)"; timeout 900 $B --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-260
echo "== bench config 5"; timeout 900 $B --n 8192 --spectrum f16 --steps 20 --warmup 5 2>/dev/null | tee $O/bench_n8192_f16.json | cut -c1-260
echo "== bench config 3 with / without the normal field"
timeout 600 $B --n 2048 --normals disp_x 2>/dev/null | tee $O/bench_n2048_normals.json | cut -c1-260
timeout 600 $B --no-cpu-baseline --n 2048 --normals height 2>/dev/null | tee $O/bench_n2048_normals_height.json | cut -c1-260
timeout 600 $B --no-cpu-baseline --n 2048 2>/dev/null | tee $O/bench_n2048.json | cut -c1-260
echo "== bench config 2 plain / batched"
timeout 600 $B --n 512 2>/dev/null | tee $O/bench_n512.json | cut -c1-260
for k in 8 16 64; do timeout 600 $B --no-cpu-baseline --n 512 --batch $k --steps 8192 --warmup 128 2>/dev/null | tee $O/bench_n512_batch$k.json | cut -c1-260; done
timeout 600 $B --no-cpu-baseline --n 512 --batch 8 --batch-tiles --steps 8192 --warmup 128 2>/dev/null | tee $O/bench_n512_batch8_tiles.json | cut -c1-260
timeout 600 $B --no-cpu-baseline --n 512 --batch 8 --normals disp_x --steps 8192 --warmup 128 2>/dev/null | tee $O/bench_n512_batch8_normals.json | cut -c1-260
timeout 600 $B --no-cpu-baseline --n 256 --batch 64 --steps 8192 --warmup 128 2>/dev/null | tee $O/bench_n256_batch64.json | cut -c1-260
timeout 600 $B --no-cpu-baseline --n 1024 --batch 16 --steps 4096 --warmup 64 2>/dev/null | tee $O/bench_n1024_batch16.json | cut -c1-260
echo "== bench 4096 with normals, 8192, 16384"
timeout 600 $B --no-cpu-baseline --n 4096 --normals disp_x 2>/dev/null | tee $O/bench_n4096_normals.json | cut -c1-260
for n in 8192 16384; do timeout 900 $B --no-cpu-baseline --n $n --steps 20 --warmup 5 --ramp-frames 20 --distribution-frames 50 2>/dev/null | tee $O/bench_n$n.json | cut -c1-260; done
cd /tmp
run_prof() {   # name, N, traffic flags, then the command
  local name=$1 n=$2 flag=$3; shift 3
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/stats -o run -- "$@" > $O/$name.stats_stdout.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$name/pmc_$c -o run -- "$@" > $O/$name.pmc_${c}_stdout.txt 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/$name > $O/$name.summary.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/make_hbm_traffic.py $O/$name $n $TAG $flag > $O/$name.hbm_traffic.txt 2>&1
  cp $GRAFT_REPO_ROOT/profiles/hbm_traffic_*.json $O/ 2>/dev/null
  echo "== $name"; grep -v "^$" $O/$name.summary.txt | cut -c1-160 | head -24; cat $O/$name.hbm_traffic.txt
}
run_prof fused_n4096 4096 - $B --no-cpu-baseline --steps 200 --warmup 5 --profile-frames 5
run_prof fused_n2048_normals 2048 normals $B --no-cpu-baseline --n 2048 --normals disp_x --steps 200 --warmup 5 --profile-frames 5
run_prof fused_n2048 2048 - $B --no-cpu-baseline --n 2048 --steps 200 --warmup 5 --profile-frames 5
run_prof fused_n4096_normals 4096 normals $B --no-cpu-baseline --n 4096 --normals disp_x --steps 100 --warmup 5 --profile-frames 5
run_prof fused_n8192_f16 8192 f16 $B --no-cpu-baseline --n 8192 --spectrum f16 --steps 60 --warmup 2 --profile-frames 2
run_prof fused_n512 512 - $B --no-cpu-baseline --n 512 --steps 200 --warmup 5 --profile-frames 3
run_prof fused_n512_batch8 512 batch8 $B --no-cpu-baseline --n 512 --batch 8 --steps 1600 --warmup 16 --profile-frames 3 --distribution-frames 20
run_prof fused_n8192 8192 - $B --no-cpu-baseline --n 8192 --steps 60 --warmup 2 --profile-frames 2
run_prof fused_n16384 16384 - $B --no-cpu-baseline --n 16384 --steps 20 --warmup 2 --profile-frames 2 --ramp-frames 10 --distribution-frames 20
R=$O/profiles_ready; mkdir -p $R
cp $O/bench.json $R/${TAG}_bench.json; for f in $O/bench_n*.json; do cp $f $R/${TAG}_$(basename $f); done
for n in fused_n512 fused_n512_batch8 fused_n2048 fused_n2048_normals fused_n4096 fused_n4096_normals fused_n8192 fused_n8192_f16 fused_n16384; do
  [ -f $O/$n.summary.txt ] && cat $O/$n.summary.txt $O/$n.hbm_traffic.txt > $R/${TAG}_${n#fused_}_rocprof_stats_and_hbm_counters.txt
done
cp $O/hbm_traffic_*.json $R/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
