#!/bin/bash
# Same-box A/B of the number of micro-batches (MicroBatchedEngine) and of the batch: bash tools/mb_ab.sh 1 2 1 2 3   or   bash tools/mb_ab.sh 2:1024 2:2048 4:2048
cd "$(dirname "$0")/.."
for v in "$@"; do
    n=${v%%:*}; b=1024; [[ "$v" == *:* ]] && b=${v##*:}
    echo "== micro-batches $n, batch $b"
    timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-traffic --cpu-docs 0 --stream-docs 0 --no-extra-rates --micro-batches "$n" --batch "$b" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],1),'docs/s  clock',round(d['clock_ghz_timed_region'],4),' per GHz',round(d['docs_per_sec_per_ghz'],1),' kv_probe',round(d['kv_probe']['docs_per_sec'],1),' ffn_up',round(d['roofline']['achieved'],1),' attn',round(d['attention_tflops'],1), 'frac', round(d['step_frac_of_ceiling'],4))" || exit 1
done
