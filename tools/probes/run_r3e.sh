R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r3e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3e_tests.log
tail -25 gpurun_out/r3e_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
