#!/usr/bin/env python3
"""dev tool: fill the @PLACEHOLDER@ numbers of DESIGN.md section 5 from profiles/r06_bench*.json (one-off, end of round 6)."""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(R + "/profiles/r06_bench.json"))
c4 = json.load(open(R + "/profiles/r06_bench_cfg4.json"))
c5 = json.load(open(R + "/profiles/r06_bench_cfg5.json"))
k1, k2 = d["north_star_kernel"], d["roofline"]
rep = {
    "STEP": "%.3f" % d["ms_per_step"], "REPEAT": "%.3f" % d["repeat_ms_per_step"]["median"], "GVOX": "%.1f" % (d["value"] / 1e3),
    "K1US": "%.0f" % k1["avg_launch_us"], "K1FRAC": "%.1f" % (100 * k1["frac"]), "K1CALL": "%.0f" % k1["whole_call_avg_us"],
    "K1VALU": "%s" % k1["valu_busy"], "K1LDS": "%s" % k1["lds_busy"], "K1HIT": "%s" % k1.get("l2_hit_rate"),
    "K1RD": "%.0f" % ((k1.get("traffic_read") or 0) / 1e6), "K1WR": "%.0f" % ((k1.get("traffic_written") or 0) / 1e6),
    "K2US": "%.0f" % k2["avg_launch_us"], "K2FRAC": "%.1f" % (100 * k2["frac"]),
    "STRESS": "%.3f" % d["stress"]["ms_per_step"], "STRESSX": "%.2f" % d["stress"]["vs_headline_ms"],
    "FRESH5": "%.3f" % d["fresh_grid"]["sigma_5"]["ms_per_step"], "FRESH10": "%.3f" % d["fresh_grid"]["sigma_10"]["ms_per_step"],
    "CFG4": "%.3f" % c4["ms_per_step"], "CFG5": "%.2f" % c5["ms_per_step"],
}
p = R + "/DESIGN.md"
s = open(p).read()
for k, v in rep.items():
    s = s.replace("@%s@" % k, v)
left = re.findall(r"@[A-Z0-9]+@", s)
open(p, "w").write(s)
print("filled; commit", d["commit"], "left:", left)
