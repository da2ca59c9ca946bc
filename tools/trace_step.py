"""Timeline of the last fused_experts call in a rocprofv3 --kernel-trace CSV: start offset, duration and queue of every
kernel between the last two `moe_align` launches (what runs beside what, and the gaps between dependent launches).
usage: python tools/trace_step.py <dir with *_kernel_trace.csv> [first kernel substring]"""
import csv
import glob
import sys

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "moe_count"
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit(f"fewer than three '{first}' launches in the trace")
a, b = starts[-3], starts[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-70:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")
    prev_end = max(prev_end, e)
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
