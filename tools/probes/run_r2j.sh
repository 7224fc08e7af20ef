R=$GRAFT_REPO_ROOT; cd $R
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
( time python bench.py > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err ) 2>&1 | tail -4; cut -c1-300 gpurun_out/r2j_bench.json
