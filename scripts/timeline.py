import csv,glob,sys
ev=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:36]))
for f in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+" ".join("%s=%s"%(k,v) for k,v in r.items() if k not in ("Start_Timestamp","End_Timestamp","Kind","Correlation_Id"))))
ev.sort()
# find the last k_eval_coop region
pat=sys.argv[2] if len(sys.argv)>2 else "eval_coop"
idx=[i for i,e in enumerate(ev) if pat in e[2]]
print(len(ev),len(idx))
if idx:
    a=idx[-40]
    t0=ev[a][0]
    for e in ev[a-2:a+26]:
        print("%10.1f us  dur %8.1f us  %s"%((e[0]-t0)/1e3,(e[1]-e[0])/1e3,e[2]))
# start-to-start periods of the selected kernel over its last run
if idx:
    st=[ev[i][0] for i in idx[len(idx)//2:]]
    per=[(b-a)/1e3 for a,b in zip(st,st[1:])]
    print("periods (us):"," ".join("%.0f"%p for p in per))
