cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 40 --warmup 10 --no-cpu --no-roofline > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv,re,glob
rows=[]
for f in glob.glob("$O/t/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"]); rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),m.group(1) if m else r["Kernel_Name"][:30]))
for f in glob.glob("$O/t/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r.get("Direction","?")[:12]))
rows.sort()
# find the last k_fast_cells and print the window around one full step
idx=[i for i,r in enumerate(rows) if r[2]=="k_fast_cells"]
i0=idx[-3]; 
# step starts a few entries before fast (uploads + resizes)
start=max(0,i0-12); t0=rows[start][0]
for s,e,n in rows[start:idx[-2]-8]:
    print("%8.1f +%6.1f us  %s" % ((s-t0)/1e3,(e-s)/1e3,n))
PY
