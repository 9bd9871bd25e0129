"""Kernel durations and the gaps between consecutive kernels of the last 40 steps of tools/gap_probe.py
(rocprofv3 --kernel-trace --output-format csv): python tools/gap_report.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mli::" in r["Kernel_Name"]]
ks = ks[-160:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (n, s, e) in enumerate(ks):
    dur[n].append((e - s) / 1e3)
    if i: gap[n].append((s - ks[i - 1][2]) / 1e3)
for n in dur:
    g = gap[n]
    print(f"{n:62s} n={len(dur[n]):3d} dur {sum(dur[n])/len(dur[n]):7.2f} us   gap before {sum(g)/max(len(g),1):6.2f} us")
span = (ks[-1][2] - ks[0][1]) / 1e3
print(f"span {span:.1f} us over {len(ks)} kernels; sum of durations {sum(sum(v) for v in dur.values()):.1f}")
