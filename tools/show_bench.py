"""Print the key figures of bench.py JSON lines."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    r = d["roofline"]
    print(path, "| ms/step", round(d["ms_per_step"], 3), "| unit_ms", round(r["kernel_avg_ms"], 3), "| merge_ms",
          round(r["merge_kernel_avg_ms"], 3), "| frac", round(r["frac"], 4), "| cand/s", f'{d["value"]:.3e}', "| parity",
          d["recall_at_400_parity"], "| fb", d["fallback_units"], "| P", d["config"]["partitions"])
