#!/bin/bash
# Evidence run for profiles/: GPU tests, smoke, sweep, bench lines (fp32 N = 4096; config 5: fp16 spectrum N = 8192),
# rocprofv3 kernel stats of the same bench commands, HBM counters (FETCH_SIZE / WRITE_SIZE in separate passes) for the
# fused frame and for the staged 8-dispatch path.   tools/gpu_evidence.sh <tag>
set -u
exec < /dev/null
TAG=${1:-ev}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | tee $O/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
echo "== sweep"; timeout 900 python tools/sweep.py 256 512 1024 2048 4096 8192 > $O/sweep.jsonl 2>&1; python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r["n"], "fused %.4f ms  %.0f fps |"%(r["fused_ms"], r["fused_fps"]), {k: round(v*1000,1) for k,v in r["fused"].items()}, "| staged %.4f ms"%r["staged_ms_total"])
PY
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench.json | cut -c1-300
echo "== bench config 5"; timeout 900 python bench.py --n 8192 --spectrum f16 --steps 20 --warmup 5 2>$O/bench_f16.err | tee $O/bench_n8192_f16.json | cut -c1-300
echo "== bench config 5, opt-in 16-bit intermediate"; timeout 900 python bench.py --no-cpu-baseline --n 8192 --spectrum f16 --intermediate bfp16 --steps 20 --warmup 5 2>/dev/null | tee $O/bench_n8192_f16_bfp16.json | cut -c1-300
echo "== bench config 2 (N = 512) and 3 (N = 2048)"; for n in 512 2048; do timeout 600 python bench.py --no-cpu-baseline --n $n --steps 200 --warmup 20 2>/dev/null | tee $O/bench_n$n.json | cut -c1-200; done
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
  echo "== $name"; grep -v "^$" $O/$name.summary.txt | cut -c1-160 | head -30
}
run_prof fused_n4096 4096 - python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 5 --profile-frames 5
run_prof fused_n8192_f16 8192 f16 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 8192 --spectrum f16 --steps 60 --warmup 2 --profile-frames 2
run_prof fused_n8192_f16_bfp16 8192 "f16 bfp16" python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 8192 --spectrum f16 --intermediate bfp16 --steps 60 --warmup 2 --profile-frames 2
run_prof staged_n4096 4096 staged python $GRAFT_REPO_ROOT/tools/staged_frames.py 4096 10
for N in 512 2048 8192; do
  run_prof fused_n$N $N - python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n $N --steps 100 --warmup 5 --profile-frames 3
done
run_prof fused_n16384 16384 - python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --n 16384 --steps 20 --warmup 2 --profile-frames 2 --ramp-frames 10 --distribution-frames 20
cd $GRAFT_REPO_ROOT
echo "== bench N = 8192 fp32 and 16384"; for n in 8192 16384; do timeout 900 python bench.py --no-cpu-baseline --n $n --steps 20 --warmup 5 --ramp-frames 20 --distribution-frames 50 2>/dev/null | tee $O/bench_n$n.json | cut -c1-200; done
echo "== race: long run of the barrier-jitter build"; timeout 1500 python tools/race_long_run.py 300 2>&1 | tail -25 | tee $O/race_long_run_jitter_build.txt
# the names profiles/ uses (copy by hand what should be judged: `cp gpurun_out/$TAG/profiles_ready/* profiles/`)
R=$O/profiles_ready; mkdir -p $R
cp $O/bench.json $R/${TAG}_bench.json; for f in $O/bench_n*.json; do cp $f $R/${TAG}_$(basename $f); done
for n in fused_n512 fused_n2048 fused_n4096 fused_n8192 fused_n8192_f16 fused_n8192_f16_bfp16 fused_n16384 staged_n4096; do
  [ -f $O/$n.summary.txt ] && cat $O/$n.summary.txt $O/$n.hbm_traffic.txt > $R/${TAG}_${n#fused_}_rocprof_stats_and_hbm_counters.txt
done
cp $O/pytest_gpu.txt $R/${TAG}_pytest_gpu.txt; cp $O/smoke.txt $R/${TAG}_smoke.txt; cp $O/sweep.jsonl $R/${TAG}_sweep.jsonl
cp $O/race_long_run_jitter_build.txt $R/${TAG}_race_long_run_jitter_build.txt; cp $O/hbm_traffic_*.json $R/ 2>/dev/null
# keep the pulled directory small: the csv traces are summarised above
find $O -name "*.csv" -size +2M -delete
