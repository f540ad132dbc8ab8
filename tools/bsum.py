import json, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(src.strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["device_ms_per_step"], d.get("whole_iteration_gbps"), d.get("ritz_backtransform", {}).get("ms"))
print({k: (v["achieved"], v["avg_us"], v["timed_launches"]) for k, v in (d.get("roofline_all") or {}).items()})
