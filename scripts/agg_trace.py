import sys,collections
t=collections.defaultdict(float);n=collections.Counter()
for l in sys.stdin:
    if "gc trace" not in l or "stream:" not in l: continue
    p=l.split("stream:")[-1].rsplit(None,2)
    k=p[0].strip();t[k]+=float(p[1]);n[k]+=1
for k in sorted(t,key=lambda k:-t[k]): print("%-30s n=%5d total %9.3f ms  avg %7.1f us"%(k,n[k],t[k],1e3*t[k]/n[k]))
