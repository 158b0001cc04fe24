import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["roofline"]["frac"])
for k,v in d["secondary"]["shape_families"].items():
    print(k, v if isinstance(v,str) else {kk:(round(vv,3) if isinstance(vv,float) else vv[:60] if isinstance(vv,str) else vv) for kk,vv in v.items()})
print([k for k,v in d["secondary"].items() if isinstance(v,dict) and "error" in v])
