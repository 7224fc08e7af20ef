R=$GRAFT_REPO_ROOT; cd $R
python tools/probes/w4_debug.py 256 256 576 1 1 2>&1 | grep -v "^[.+#]*$" | tail -4
python tools/probes/w4_debug.py 76800 256 576 1 1 2>&1 | grep -v "^[.+#]*$" | tail -4
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "w4" > gpurun_out/r2c_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2c_pytest.log
