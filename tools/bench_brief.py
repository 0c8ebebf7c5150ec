import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]/1e6,2), "rows", round(d["roofline"]["kernel_us"],2), "eval", {k:(round(v,2) if not isinstance(v,list) else [round(x,1) for x in v]) for k,v in d["eval_us"].items()})
