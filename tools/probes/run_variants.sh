# usage: run_variants.sh "<gemm_bench args>" variant...   (same box, back to back, two passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="$1"; shift
for pass in 1 2; do
for v in "$@"; do
  echo "== $v (pass $pass)"
  MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$v.so timeout 300 python $R/tools/gemm_bench.py $ARGS 2>&1 | grep -v amdgpu.ids
done
done
