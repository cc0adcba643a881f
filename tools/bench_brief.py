"""Dev tool: the figures of a bench line worth a glance.  usage: bench_brief.py <file with the JSON line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d.get("solves", {})
print("evals/s", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("solves/s", s.get("value"), "s", s.get("seconds"), "4096:", (d.get("solves_4096_scenarios") or {}).get("value"))
print("tick", {k: v for k, v in (d.get("single_tick_default_yaml") or {}).items() if k.endswith("ms_per_tick")})
