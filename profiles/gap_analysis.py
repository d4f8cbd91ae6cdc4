"""GPU idle time between kernels of a rocprofv3 kernel trace: python profiles/gap_analysis.py <kernel_trace.csv> [min_gap_us]
Groups the gaps by (kernel before, kernel after) and prints the largest contributors; the total is the time the GPU waited for the host
(descriptor uploads, D2H read-backs of ranks / bond dimensions, Python) inside the traced interval."""
import csv, re, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("tnqs::", "")))
rows.sort()
ming = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 5e3
gaps = defaultdict(lambda: [0, 0.0]); tot = 0.0; busy = 0.0; end = rows[0][0]
for s, e, n in rows:
    if s > end:
        g = s - end
        if g >= ming and g < 5e7:          # ignore the pauses between steps / set-up
            gaps[(prev, n)][0] += 1; gaps[(prev, n)][1] += g; tot += g
    busy += max(0, e - max(s, end)); end = max(end, e); prev = n
print(f"kernels {len(rows)}  busy {busy/1e6:.1f} ms  gaps >= {ming/1e3:.0f} us: {tot/1e6:.1f} ms")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:8.2f} ms  {c:5d} x {t/c/1e3:7.1f} us   {a[:48]:48s} -> {b[:48]}")
