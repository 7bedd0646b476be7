import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line); print(d["value"], d["ms_per_step"], {k:(v["us_per_step"]) for k,v in d["kernels"].items() if k.startswith("render")})
