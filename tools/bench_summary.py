import sys, json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith("{"): continue
    j=json.loads(l)
    e=j.get("eighth_gpu") or {}
    print(j["config"]["workload"][:28], j["value"], j["roofline"]["frac"], j["roofline"].get("kernel"), e.get("value"), e.get("frac_of_measured_copy"))
