"""Dev tool: overlap summary of the kernel trace written by rocprofv3 for tools/h100_trace.py."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    tag = "E" if "expand_kernel" in name else ("R" if "prefix_kernel" in name else None)
    if tag: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag, r.get("Queue_Id", "?")))
ev.sort()
ev = ev[-32:]          # the last launch: 16 chunks
t0 = ev[0][0]
for s, e, tag, q in ev:
    print(f"{tag} q={q} start {1e-6*(s-t0):8.3f} ms  end {1e-6*(e-t0):8.3f} ms  dur {1e-6*(e-s):6.3f}")
print("span ms", 1e-6 * (max(e for s, e, *_ in ev) - t0))
