R=$GRAFT_REPO_ROOT; cd $R
echo "== per-CU 1 vs 2, shipped lib"
for pc in 1 2; do MSCLIP_PP2_PER_CU=$pc python tools/gemm_bench.py --tiles 8 2>&1 | grep -E "^(qkv|out|fc|proj|qkvnb)"; done
for l in nodma noepi nodmanoepi prio; do echo "== $l"; MSCLIP_HIP_LIB=$R/tools/probes/libgemm_$l.so python tools/gemm_bench.py --tiles 4 8 2>&1 | grep -E "^(qkv |out|fc |proj|qkvnb)"; done
echo "== nodmanoepi per-CU 1"; MSCLIP_PP2_PER_CU=1 MSCLIP_HIP_LIB=$R/tools/probes/libgemm_nodmanoepi.so python tools/gemm_bench.py --tiles 8 2>&1 | grep -E "^(qkv |out|fc |proj|qkvnb)"
echo "== trace qkvnb"; MSCLIP_HIP_LIB=$R/tools/probes/libgemm_trace.so python tools/probes/pp2_trace.py qkvnb 2>&1 | tail -12
echo "== trace qkv delay 0"; MSCLIP_PP2_DELAY=0 MSCLIP_HIP_LIB=$R/tools/probes/libgemm_trace.so python tools/probes/pp2_trace.py qkv 2>&1 | tail -12
