cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 60 --warmup 20 --no-cpu --no-roofline > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv,re,glob
rows=[]
for f in glob.glob("$O/t/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"]); rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),m.group(1) if m else r["Kernel_Name"][:30], r.get("Queue_Id","?")))
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2]=="k_resolve"]
# window: from the 30th resolve to the 33rd (three timesteps in steady state, before the serial-latency section)
i0=idx[30]; i1=idx[33]
t0=rows[i0][0]
for s,e,n,q in rows[i0:i1+1]:
    print("%8.1f +%6.1f us  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,q,n))
PY
