R=$GRAFT_REPO_ROOT; cd $R
python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline 2>/dev/null | tail -1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -o run -- python $R/bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-probe > $R/gpurun_out/prof_b16.log 2>&1
