#!/bin/bash
# Same-box A/B of the split GEMM's work-queue ORDER for the wide layer GEMMs (N >= 2304: Q|K|V, FFN-up), VERDICT r05 item 5: whole bench steps on
# the DIAGNOSTIC library with MMEE_GEMM_ORDER = -1 (shipped: groups of 8 M-tiles, M fastest), 3 (W-stationary: an XCD keeps 4 N-tiles and
# sweeps its M-groups) and 4 (A-panel: one 256-row A panel through all its N-tiles).  Prints docs/s, the clock held, the FFN-up kernel's rate
# and its HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE from the bench's own rocprofv3 --pmc child passes, which inherit the switch).
# Usage (GPU box): bash tools/gemm_order_ab.sh -1 3 4 -1 3 4
cd "$(dirname "$0")/.."
export MMEE_LIB=$PWD/multi-modal-early-exit_amd/libmmee_hip_diag.so
for o in "$@"; do
    echo "== MMEE_GEMM_ORDER=$o"
    MMEE_GEMM_ORDER=$o timeout -k 10 400 python bench.py --steps 6 --warmup 2 --cpu-docs 0 --stream-docs 0 --no-extra-rates --no-small-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
td=r.get('traffic_detail',{})
print('docs/s',round(d['value'],1),' clock',round(d['clock_ghz_timed_region'],4),' per GHz',round(d['docs_per_sec_per_ghz'],1),' frac',round(d['step_frac_of_ceiling'],4),
      ' gemm class',round(d['gemm_class_tflops'],1),' ffn_up TF',round(r['achieved'],1),' ffn_up fetch GB/launch',round(2*td.get('FETCH_SIZE_KiB_raw_per_launch',0)*1024/1e9,3),
      ' write GB',round(td.get('WRITE_SIZE_KiB_per_launch',0)*1024/1e9,3),' prof clock',r.get('clock_ghz'),' mfma_busy',r.get('mfma_busy'),
      ' shares',{k:v for k,v in d['kernel_time_share'].items() if k.startswith('gemm_')})" || exit 1
done
