import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g ms/step %.4f frac %.4f parity %s" % (l["value"], l["ms_per_step"], l["roofline"]["frac"], l.get("parity_ok")))
print(json.dumps(l.get("secondary"), indent=1)[:4000])
print(l.get("cpu_baseline"))
