"""Print the few numbers of a bench.py JSON line that rounds are compared by (stdin or a file of lines)."""
import json
import sys

for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    if "error" in d:
        print("error:", d["error"])
        continue
    fw = d.get("fixed_work_rates", {})
    print(f"docs/s {d['value']:.1f}  ms/step {d['ms_per_step']:.2f}  clock {d.get('clock_ghz_timed_region') or 0:.4f} GHz  per GHz {d.get('docs_per_sec_per_ghz') or 0:.1f}  "
          f"mean exit layer {d.get('mean_exit_layer')}  step_frac {d.get('step_frac_of_ceiling') or 0:.4f}  no-exit {fw.get('no_exit', {}).get('docs_per_sec', 0):.1f} "
          f"(frac {fw.get('no_exit', {}).get('step_frac_of_ceiling', 0):.4f}, clock {fw.get('no_exit', {}).get('clock_ghz') or 0:.4f})  "
          f"ffn_up {d.get('roofline', {}).get('achieved', 0):.1f} TF (frac {d.get('roofline', {}).get('frac', 0):.4f})  attn {d.get('attention_tflops') or 0:.1f}  "
          f"parity {d.get('parity_vs_cpu_sample')}")
