R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "golden or c4_per_rank or taps" 2>&1 | grep "observed margins\|C4 per-rank\|worst taps\|passed\|failed" | cut -c1-300
