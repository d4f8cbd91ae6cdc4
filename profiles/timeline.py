"""Timeline of the last layer of a rocprofv3 kernel trace: python profiles/timeline.py <kernel_trace.csv> <layers in the trace> [max_lines]
One line per kernel launch: start and end offset (us), duration (us), idle gap before it (us), workgroups, queue, name.  The dependent chains of small
kernels between the heavy passes of a gate batch are read off this listing."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("tnqs::", "")
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, wg, r.get("Queue_Id", "?")))
rows.sort()
nl = int(sys.argv[2]); per = len(rows) // nl
last = rows[len(rows) - per:]
t0 = last[0][0]; end = t0; busy = 0
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
print(f"{len(rows)} launches in the trace, {per} per layer; last layer: {(last[-1][1] - t0) / 1e3:.1f} us")
for i, (s, e, n, wg, q) in enumerate(last):
    if i < lim: print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {max(0, s - end) / 1e3:7.1f} {wg:7d} q{q} {n[:70]}")
    busy += max(0, e - max(s, end)); end = max(end, e)
print(f"busy {busy / 1e3:.1f} us of {(end - t0) / 1e3:.1f} us")
