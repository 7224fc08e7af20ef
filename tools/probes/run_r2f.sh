R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2f_pytest.log | cut -c1-220
