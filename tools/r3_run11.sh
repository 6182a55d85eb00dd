cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/r3_bt; mkdir -p gpurun_out/r3_bt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3_bt/kt -o run -- python3 tools/batch_trace.py ${INFL:-8} > gpurun_out/r3_bt/log.txt 2>&1
tail -2 gpurun_out/r3_bt/log.txt
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r3_bt/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(len(rows), list(rows[0].keys()))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"],r["Kernel_Name"]) for r in rows]
ev.sort()
# take the last 40% of the run (the timed batch)
t_end=max(e[1] for e in ev); t_beg=min(e[0] for e in ev)
cut=t_end-130_000_000   # last 130 ms
sel=[e for e in ev if e[0]>=cut]
busy=sum(e[1]-e[0] for e in sel)
# union
pts=[]
for s,e,_,_ in sel: pts.append((s,1)); pts.append((e,-1))
pts.sort()
cur=0; last=None; union=0; hist=collections.Counter()
for t,d in pts:
    if last is not None and cur>0: union+=t-last; hist[min(cur,16)]+=t-last
    cur+=d; last=t
span=max(e[1] for e in sel)-min(e[0] for e in sel)
print("kernels %d  span %.1f ms  sum of durations %.1f ms  union busy %.1f ms  avg concurrency %.2f" % (len(sel), span/1e6, busy/1e6, union/1e6, busy/max(union,1)))
print("time by concurrency level (ms):", {k: round(v/1e6,1) for k,v in sorted(hist.items())})
q=collections.Counter(e[2] for e in sel); print("queues:", dict(q))
byk=collections.defaultdict(lambda:[0,0])
for s,e,_,n in sel: byk[n[:50]][0]+=1; byk[n[:50]][1]+=e-s
for n,(c,t) in sorted(byk.items(), key=lambda x:-x[1][1])[:12]: print("  %-52s %5d  %.1f ms  avg %.1f us" % (n,c,t/1e6,t/c/1e3))
PY
find gpurun_out/r3_bt -name "*kernel_trace.csv" -delete
