import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(p["value"], p["ms_per_step"], p["fp32_mode"])
for nm in ("parity_mode_f16x3","fp16_mode"):
    q=p[nm]; r=q.pop("roofline"); print(nm, q); ks=r.pop("kernels"); print({k:v for k,v in r.items() if k in ("frac","by_class","all_conv_ms_per_step","all_kernels_ms_per_step")})
for k in ks[:14]: print(k)
