"""Print the key figures of bench.py JSON lines."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    r = d.get("roofline") or {}
    print(path, "| ms/step", round(d["ms_per_step"], 3), "| unit_ms", round(r.get("kernel_avg_ms") or 0, 3), "| merge_ms",
          round(r.get("merge_kernel_avg_ms") or 0, 3), "| frac", round(r.get("frac") or 0, 4), "| cand/s", f'{d["value"]:.3e}', "| parity",
          d.get("recall_at_400_parity"), "| fb", d.get("fallback_units"), "| P", d["config"].get("partitions"), "| sharding", d["config"].get("sharding"))
