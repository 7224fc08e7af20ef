R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r5e_trace -o run -- python $R/bench.py --train --bn frozen --no-cpu-baseline --no-probe --no-pmc --steps 6 --warmup 3 > $O/r5e_trace.log 2>&1
cd $R
T=$(find $O/r5e_trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T --top 45 --gaps 15 > $O/r5e_train_timeline_frozen.txt 2>&1
rm -rf $O/r5e_trace
head -120 $O/r5e_train_timeline_frozen.txt
