cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5bg; mkdir -p $O
export TNQS_BENCH_NOPROF=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/ro -- python $R/bench.py --bp-order reference --steps 1 --warmup 1 --no-cpu-baseline > $O/ro.log 2>&1
f=$(ls /tmp/ro/*/*kernel_trace.csv | head -1)
python - <<PY
import csv, re, collections
rows=[]
for r in csv.DictReader(open("$f")):
    nm=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("tnqs::","")
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),nm,int(r.get("Grid_Size_X",r.get("Grid_Size",0)) or 0)//max(1,int(r.get("Workgroup_Size_X",r.get("Workgroup_Size",1)) or 1))))
rows.sort()
n=len(rows); last=rows[n//2:]   # second (timed) step roughly
t0=last[0][0]
dur=collections.Counter(); cnt=collections.Counter(); gaps=0; end=t0
for s,e,nm,wg in last:
    dur[nm]+=e-s; cnt[nm]+=1; gaps+=max(0,s-end); end=max(end,e)
print("launches", len(last), "span ms", (end-t0)/1e6, "gaps ms", gaps/1e6)
for nm,v in dur.most_common(14): print(f"{v/1e6:8.2f} ms {cnt[nm]:5d} avg {v/cnt[nm]/1e3:7.1f} us  {nm[:60]}")
# a window of 60 launches in the middle of a BP update
mid=len(last)//3
for s,e,nm,wg in last[mid:mid+45]: print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:7.1f} wg {wg:5d} {nm[:50]}")
PY
