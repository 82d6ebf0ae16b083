#!/bin/bash
# round 6: when do the long (deep) kernels of ssa23's garbling pass run, and how many at a time?
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06ssa; mkdir -p $OUT
cd /tmp; rm -rf /tmp/kt_ssa
rocprofv3 --kernel-trace -f csv -d /tmp/kt_ssa -o kt -- python $REPO/scripts/bench_stream.py ${1:-ssa23:64} > $OUT/bench.log 2>&1
python $REPO/scripts/lanes_dump.py /tmp/kt_ssa garble > $OUT/garble_kernels.csv
python - $OUT/garble_kernels.csv <<'PY' | tee $OUT/deep_overlap.txt
import sys
ev=[l.strip().split(",", 4) for l in open(sys.argv[1])]
ev=[(float(a),float(b),q,int(w),n) for a,b,q,w,n in ev]
span=max(a+b for a,b,*_ in ev)
longk=[e for e in ev if e[1]>=400]
print("garble pass: %d kernels over %.1f ms; %d kernels of >= 0.4 ms, together %.1f ms" % (len(ev), span/1e3, len(longk), sum(e[1] for e in longk)/1e3))
# concurrency of long kernels over time
pts=sorted([(a,1) for a,b,*_ in longk]+[(a+b,-1) for a,b,*_ in longk])
cur=0; last=0; hist={}
for t,d in pts:
    hist[cur]=hist.get(cur,0)+(t-last); last=t; cur+=d
print("time with k long kernels running: " + ", ".join("%d: %.1f ms" % (k, v/1e3) for k,v in sorted(hist.items())))
pts=sorted([(a,1) for a,b,*_ in ev]+[(a+b,-1) for a,b,*_ in ev])
cur=0; last=0; hist={}
for t,d in pts:
    hist[cur]=hist.get(cur,0)+(t-last); last=t; cur+=d
print("time with k pass kernels running (any length): " + ", ".join("%d: %.1f ms" % (k, v/1e3) for k,v in sorted(hist.items())))
print("the 12 longest kernels: start ms, duration ms, queue")
for e in sorted(longk, key=lambda e:-e[1])[:12]: print("   %.2f %.2f q%s" % (e[0]/1e3, e[1]/1e3, e[2]))
PY
tail -n 2 $OUT/bench.log | cut -c1-300
