import csv, sys
rows=[]
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
acc=[(s,e) for s,e,n in rows if n.startswith("k_accum_affine")]
last=acc[-6:]
print("accum_affine us:", [round((e-s)/1e3,1) for s,e in last])
for nm in ("k_accum_jac","k_accum_jac_q4","void k_bucket_chunks<4>","k_plan_apply","k_final_sum"):
    xs=[(s,e) for s,e,n in rows if n==nm and s>=last[0][0]]
    print(nm, len(xs), round(sum(e-s for s,e in xs)/1e3,1))
