#!/bin/bash
# rocprofv3 stats + HBM counters of the staged 8-dispatch path alone:  tools/gpu_evidence_staged.sh <tag> [N]
set -u
exec < /dev/null
TAG=${1:-staged}; N=${2:-4096}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
name=staged_n$N
CMD="python $GRAFT_REPO_ROOT/tools/staged_frames.py $N 10"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/stats -o run -- $CMD > $O/$name.stats_stdout.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$name/pmc_$c -o run -- $CMD > $O/$name.pmc_${c}_stdout.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $O/$name > $O/$name.summary.txt 2>&1
python $GRAFT_REPO_ROOT/tools/make_hbm_traffic.py $O/$name $N $TAG staged > $O/$name.hbm_traffic.txt 2>&1
cp $GRAFT_REPO_ROOT/profiles/hbm_traffic_staged_n$N.json $O/ 2>/dev/null
grep -v "^$" $O/$name.summary.txt | cut -c1-170 | tail -14
find $O -name "*.csv" -size +2M -delete
