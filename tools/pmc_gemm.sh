# usage: bash tools/pmc_gemm.sh <tile> <shape> <tag>
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
T=${1:-2}; S=${2:-proj}; TAG=${3:-x}
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -o run -- python $R/tools/gemm_pmc.py $T $S > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1; grep -E "^(qkv|proj|fc|out)" $R/gpurun_out/pmc_${TAG}_$1.log; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
run sq3 "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_IFETCH"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
