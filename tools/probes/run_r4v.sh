R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o run -- python $R/bench.py --train --bn frozen --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-pmc > $O/r4v.log 2>&1
tail -1 $O/r4v.log | cut -c1-200
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-200
