#!/bin/bash
# A/B of pass-2 variants + HBM counters.  Outputs under gpurun_out/r2/.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest gpu (thin default)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
for v in thin fat; do
  echo "== sweep $v"; OCEAN_PASS2=$v timeout 600 python tools/sweep.py 1024 2048 4096 8192 2>&1 | tee $O/sweep_$v.jsonl
done
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tee $O/bench.json
cd /tmp
echo "== rocprof list"; timeout 120 rocprofv3 -L > $O/counters_avail.txt 2>&1; grep -c . $O/counters_avail.txt
echo "== rocprof kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $O/rocprof_stats_stdout.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  for v in thin fat; do
    echo "== pmc $c $v"
    OCEAN_PASS2=$v timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_$v -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --profile-frames 2 > $O/pmc_${c}_${v}_stdout.txt 2>&1
  done
  echo "== pmc $c membench"
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_membench -o run -- $GRAFT_REPO_ROOT/tools/membench > $O/pmc_${c}_membench_stdout.txt 2>&1
done
echo "== pmc SQ"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --profile-frames 2 > $O/pmc_sq_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $O > $O/summary.txt 2>&1; head -150 $O/summary.txt
du -sh $O
