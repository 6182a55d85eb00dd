# PMC counters of the quotient kernels of constraint programs (tools/airq_fixed.py): separate counter-only runs
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/pmc_terms
rm -rf $P; mkdir -p $P
for c in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $P/$c -o run -- python3 tools/airq_fixed.py > $P/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
P="gpurun_out/pmc_terms"
tab=collections.defaultdict(dict)
for d in sorted(os.listdir(P)):
    for f in glob.glob(P+"/"+d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("zk::","")
            if "quotient_air" not in k: continue
            acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in acc.items(): tab[k][d]=v/max(n,1)
cols=sorted({c for v in tab.values() for c in v})
with open(P+"/summary.md","w") as o:
    o.write("| kernel | "+" | ".join(cols)+" |\n|---|"+"---|"*len(cols)+"\n")
    for k,v in sorted(tab.items()): o.write("| `%s` | " % k + " | ".join("%.4g" % v.get(c,float('nan')) for c in cols)+" |\n")
print(open(P+"/summary.md").read())
PY
find $P -name "*counter_collection.csv" -size +2000k -delete
