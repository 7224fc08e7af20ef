#!/bin/bash
# fused front kernels: parity tests + timings against the unfused chain on the same box (8-wave and 4-wave forms)
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_conv or fused_with_next" 2>&1 | tail -5
timeout 600 python tools/front_bench.py 2>&1 | tail -4
