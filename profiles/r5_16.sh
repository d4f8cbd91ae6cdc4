O=gpurun_out/r5v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1 3 4; do
  TNQS_X3_MODE=$m rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/prof_$m -o out --output-format csv -- python $GRAFT_REPO_ROOT/profiles/x3_clock.py > /tmp/log_$m.txt 2>&1
  echo "== mode $m"; tail -1 /tmp/log_$m.txt
  python - <<PY
import csv, glob
for f in glob.glob("/tmp/prof_$m/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[-2:]:
        print({k: r[k] for k in r if k in ("Kernel_Name","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp")})
for f in glob.glob("/tmp/prof_$m/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gram2" in r["Kernel_Name"]]
    for r in rows[-2:]:
        print("dur_ns", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/clock.txt
