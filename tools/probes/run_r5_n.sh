R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -4
for bn in frozen batch; do
python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$bn', r['ms_per_step'], r['value'])"
done | tee $O/r5n_train_lines.txt
cd /tmp; export TMPDIR=/tmp
for bn in frozen batch; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5n_prof_$bn -o run -- python $R/bench.py --train --bn $bn --no-cpu-baseline --no-probe --no-pmc --steps 8 --warmup 3 > $O/r5n_prof_$bn.log 2>&1
T=$(find $O/r5n_prof_$bn -name "*kernel_trace.csv" | head -1)
(cd $R; python tools/timeline.py $T --top 40 --gaps 12 > $O/r5n_timeline_$bn.txt 2>&1)
find $O/r5n_prof_$bn -name "*kernel_trace.csv" -delete; find $O/r5n_prof_$bn -name "*agent_info.csv" -delete
head -3 $O/r5n_timeline_$bn.txt
done
