import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(p["value"], p["ms_per_step"], p["fp32_mode"])
q=p["parity_mode_f16x3"]; r=q.pop("roofline"); print(q); ks=r.pop("kernels"); print(r)
for k in ks[:14]: print(k)
