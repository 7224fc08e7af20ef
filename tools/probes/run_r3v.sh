#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python tools/probes/host_time.py frozen 512 2>/dev/null | tail -1; }
run A=1
run HSA_ENABLE_INTERRUPT=0
run ROC_AQL_QUEUE_SIZE=65536
run GPU_MAX_HW_QUEUES=8
run ROC_SIGNAL_POOL_SIZE=2048
run HIP_FORCE_DEV_KERNARG=0
run ROC_ACTIVE_WAIT_TIMEOUT=200
run AMD_DIRECT_DISPATCH=0
