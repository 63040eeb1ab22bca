"""Per-step kernel table of a training-step trace (rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/pgpr_train_one.py):
    python tools/sum_train_trace.py <dir> [rows] [marker kernel substring, one launch per step]"""
import csv, collections, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
marker = sys.argv[3] if len(sys.argv) > 3 else "reinforce"
key=[i for i,nm in enumerate(names) if marker in nm]
i0,i1=key[-11],key[-1]
seg=rows[i0:i1]
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    k=r["Kernel_Name"].split("(")[0][:60] if "at::native" not in r["Kernel_Name"] else r["Kernel_Name"][:230]
    agg[k][0]+=1; agg[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
steps=10
tot=sum(v[1] for v in agg.values())
print("launches/step", len(seg)/steps, "busy us/step", round(tot/steps/1e3,1), "wall us/step", round((int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/steps/1e3,1))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print(f"{v[0]/steps:6.1f} {v[1]/steps/1e3:8.1f} us  {k}")
