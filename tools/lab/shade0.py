import csv,glob,sys,re
f=glob.glob(sys.argv[1]+"/**/*_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
names=[("raygen" if "k_raygen" in r["Kernel_Name"] else "shade" if "k_shade" in r["Kernel_Name"] else "x",(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows]
out=[]
for i,(n,d) in enumerate(names):
    if n=="raygen":
        for m,d2 in names[i+1:]:
            if m=="shade": out.append(d2); break
print("bounce-0 k_shade durations (us):",[round(x) for x in out])
