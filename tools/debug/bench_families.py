import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["final_loss"], {k:(round(v["ms_per_step"],2), round(v.get("tflops",0),1)) for k,v in r["families"].items()})
