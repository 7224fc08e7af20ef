R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/lk -o run -- python $R/bench.py --train --bn frozen --steps 3 --warmup 2 --no-cpu-baseline --no-probe --no-pmc > $O/r4y.log 2>&1
f=$(find /tmp/lk -name "*kernel_trace.csv" | head -1)
head -1 $f | cut -c1-400
python $R/tools/probes/long_kernels.py $f 200 | tee $O/r4y_long.txt
