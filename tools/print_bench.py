"""Prints the headline of bench.py JSON lines: value, ms per step, steady state, kernel times alone."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    alone = d["roofline"].get("dominant_kernel_alone", {}).get("kernel_ms_all")
    print(f.split("/")[-1], "%.3f M" % (d["value"] / 1e6), "ms/step %.3f" % d["ms_per_step"],
          "steady %.3f M" % (d["steady_state"]["value"] / 1e6) if "steady_state" in d else "",
          {k: round(v, 3) for k, v in (alone or d["kernel_ms"]).items()})
