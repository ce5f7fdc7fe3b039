import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","ms_per_step","n_gpus","steps","warmup")})
print("roofline",d["roofline"]["frac"],d["roofline"].get("traffic"))
ek=d.get("extra_keys",d)
for k in ("config5_blocks","full_xengine_concurrent","sync_per_call","corr_block","sustained","fits_in_driver_run"):
    v=(ek.get(k) if isinstance(ek,dict) else None) or d.get(k)
    print(k, json.dumps(v)[:260] if v else None)
