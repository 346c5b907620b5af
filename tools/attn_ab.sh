#!/bin/bash
# Same-box A/B of two builds of libcor_amd.so on the attention micro-benchmark (alternating processes).
# Usage (GPU box): bash tools/attn_ab.sh OUT.jsonl BASE.so [ROUNDS]
out=$1; base=$2; rounds=${3:-3}
: > "$out"
for r in $(seq 1 "$rounds"); do
  for lib in "$base" ""; do
    COR_AMD_LIB=$lib python3 tools/attn_bench.py 32 2>/dev/null | python3 -c "
import json, sys
lib = sys.argv[1] or 'new'
for l in sys.stdin:
    d = json.loads(l); d['lib'] = lib; print(json.dumps(d))" "$lib" >> "$out" || exit 1
  done
done
python3 - "$out" <<'PY'
import json, sys, collections
acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); acc[(d['lib'], d['window'], d['variant'], d['q_prescale'])].append(round(d['ms'], 4))
for k in sorted(acc): print(k, acc[k])
PY
