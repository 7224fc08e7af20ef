R=$GRAFT_REPO_ROOT
cd $R; python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 > $R/gpurun_out/bench_now.json; cat $R/gpurun_out/bench_now.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_now -o run -- python $R/bench.py --no-cpu-baseline --no-probe > $R/gpurun_out/prof_now.log 2>&1
tail -1 $R/gpurun_out/prof_now.log
