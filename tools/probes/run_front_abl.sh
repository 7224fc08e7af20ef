#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$v.so timeout 300 python $R/tools/front_bench.py 2>&1 | grep -v amdgpu.ids
done
