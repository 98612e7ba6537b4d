#!/usr/bin/env python
"""One line per bench run: the inference kernel's time and fraction of the matrix peak (reads bench.py's JSON on stdin)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d["roofline"].get("predict") or {}
print("cells/s %.0f  step %.4f ms  predict %.2f ms  frac %.3f" % (d["value"], d["config"].get("lane_step_ms", 0), p.get("ms", 0), p.get("frac", 0)))
