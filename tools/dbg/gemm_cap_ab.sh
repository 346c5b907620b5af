set -e
for r in 1 2; do
for cap in 0 128 192 224; do
  if [ $cap = 0 ]; then unset COR_GEMM_MAX_BLOCKS; else export COR_GEMM_MAX_BLOCKS=$cap; fi
  timeout -k 10 280 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cap $cap', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])"
done; done
