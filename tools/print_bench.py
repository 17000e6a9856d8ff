"""One-line digest of a bench.py JSON line read from stdin (ms per step, RTF, roofline, unfused statistics passes)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("%.2f ms/step, %.0fx, family %.1f TFLOP/s (frac %.4f), in_stats %s launches/step" % (
    d["ms_per_step"], d["value"], r.get("achieved", 0.0), r.get("frac", 0.0), (d.get("in_stats") or {}).get("launches_per_step")))
