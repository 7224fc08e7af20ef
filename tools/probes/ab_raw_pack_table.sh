set -e
F="--train --bn batch --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels"
for r in 1 2; do
  for v in 0 1; do
    MSCLIP_RAW_PACK_TABLE=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('raw_pack_table=$v', d['ms_per_step'], d['value'])"
  done
done
