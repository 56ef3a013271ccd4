"""Gaps between consecutive step-kernel launches of a lockstep run, from a rocprofv3 --kernel-trace CSV: usage launch_gaps.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
steps = [r for r in rows if "step_kernel" in r[2]]
gaps = [(steps[i + 1][0] - steps[i][1]) / 1e3 for i in range(len(steps) - 1)]
durs = [(e - s) / 1e3 for s, e, _ in steps]
print("launches", len(steps), "mean kernel us %.1f" % (sum(durs[-20:]) / 20), "gaps (us) of the last 20:", [round(g, 1) for g in gaps[-20:]])
