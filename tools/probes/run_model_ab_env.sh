#!/bin/bash
# same-box A/B of an environment knob: run_model_ab_env.sh VAR  -> bench with VAR unset / VAR=1, two passes
cd "$(dirname "$0")/../.."
for pass in 1 2; do
  echo "== default (pass $pass)"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  for v in "$@"; do
    echo "== $v=1 (pass $pass)"; env $v=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  done
done
