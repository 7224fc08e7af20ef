R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/ha -o run -- python $R/bench.py --train --bn frozen --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-pmc > $O/r4o.log 2>&1
ls /tmp/ha/* | head
k=$(find /tmp/ha -name "*kernel_trace.csv" | head -1); a=$(find /tmp/ha -name "*hip_api_trace.csv" | head -1)
head -2 $a
python $R/tools/probes/gap_hosts.py $k $a > $O/r4o_gap_hosts.txt 2>&1
head -60 $O/r4o_gap_hosts.txt
