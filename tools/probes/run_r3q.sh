#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "adamw or optimizer or training_loop or checkpoint or odd_batches" 2>&1 | tail -3
bash tools/probes/run_r3o.sh
