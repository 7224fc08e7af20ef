set -e
F="--train --bn batch --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels"
for r in 1 2; do
  for v in 0 1; do
    MSCLIP_ADAPTER_BN_VIEWS=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapter_bn_views=$v', d['ms_per_step'], d['value'])"
  done
done
