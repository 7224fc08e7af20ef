#!/bin/bash
# round 4: producer epilogue with non-temporal fp32 residual loads and / or stores against the shipped form
R=$GRAFT_REPO_ROOT; cd $R
bash tools/probes/run_bench_ab.sh shipped ntx ntst ntld
bash tools/probes/run_bench_ab.sh shipped ntx ntst ntld
