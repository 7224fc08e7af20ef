# HBM traffic counters of the benchmark step (separate passes, as MI355X_MICROARCH.md prescribes):
#   bash tools/pmc_bench.sh <tag> [bench args...]      -> gpurun_out/pmcb_<tag>_{FETCH_SIZE,WRITE_SIZE}/
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
TAG=${1:-x}; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcb_${TAG}_$c -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-pmc "$@" > $R/gpurun_out/pmcb_${TAG}_$c.log 2>&1
  tail -1 $R/gpurun_out/pmcb_${TAG}_$c.log | cut -c1-100
done
