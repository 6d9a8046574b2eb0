#!/bin/bash
# final_profile.sh TAG: the numbers profiles/<TAG>_* are made of, on ONE box: default bench line, cfg4 and cfg5 lines, rocprofv3
# kernel trace of six steps with its per-step shares by family and by stream.  Run through gpurun from the repository root.
TAG=${1:-final}
set -o pipefail
O=gpurun_out
timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
echo "default done"; tail -c 600 $O/${TAG}_bench.json | head -c 300; echo
timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline > $O/${TAG}_bench_cfg4.json 2> $O/${TAG}_cfg4.err || exit 1
echo "cfg4 done"
timeout -k 10 400 python bench.py --config cfg5 --scenarios 5120 --no-cpu-baseline > $O/${TAG}_bench_cfg5.json 2> $O/${TAG}_cfg5.err || exit 1
echo "cfg5 done"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$O/${TAG}_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-exact-pass --no-isolated-pass > $GRAFT_REPO_ROOT/$O/${TAG}_prof_bench.json 2> $GRAFT_REPO_ROOT/$O/${TAG}_prof.err || exit 1
cd $GRAFT_REPO_ROOT
python scripts/perstep.py $O/${TAG}_prof > $O/${TAG}_perstep.txt && python scripts/queue_budget.py $O/${TAG}_prof >> $O/${TAG}_perstep.txt
cp $(ls $O/${TAG}_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_prof
head -30 $O/${TAG}_perstep.txt
