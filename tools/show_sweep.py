import json,sys
d=json.load(open(sys.argv[1]))["coalesce_by_batch_rows"]
for k,v in d["points"].items():
    if "error" in v: print(k, v); continue
    print(f"{k:18s} ms={v['ms']:7.3f} frac={v['frac']:.3f} pushes={v['pushes']:3d} out_batches={v['output_batches']:6d} kern_ms={sum(v['kernel_ms'].values()):.2f} launches={sum(v['kernel_launches'].values())}")
print(d["cpu_1core_Mrows_per_s_by_batch_rows"], d["cpu_cores_granted"])
