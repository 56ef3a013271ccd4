"""How the step kernels of a grouped lockstep run share the device, from a rocprofv3 --kernel-trace CSV (no counters: kernels are not
serialised): for the last `n` periods, the time during which 0 / 1 / 2 / ... step kernels are executing, per stream the kernel durations
and the gaps between a stream's consecutive step kernels (order kernel + plant kernel + launch latency of a period).
usage: groups_overlap.py <dir> [n]"""
import csv, glob, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id")))
steps = sorted(r for r in rows if "step_kernel" in r[2])
streams = sorted({r[3] for r in steps})
last = steps[-n * len(streams):]
t0, t1 = min(r[0] for r in last), max(r[1] for r in last)
ev = sorted([(r[0], 1) for r in last] + [(r[1], -1) for r in last])
active, prev, hist = 0, t0, {}
for t, d in ev:
    hist[active] = hist.get(active, 0) + (t - prev)
    active, prev = active + d, t
tot = t1 - t0
print(f"step kernels of the last {n} periods per stream: {len(last)} launches on {len(streams)} streams over {tot / 1e6:.3f} ms")
print("share of that time with k step kernels executing: " + ", ".join(f"k={k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))
for s in streams:
    mine = [r for r in last if r[3] == s]
    durs = [(e - b) / 1e3 for b, e, _, _ in mine]
    gaps = [(mine[i + 1][0] - mine[i][1]) / 1e3 for i in range(len(mine) - 1)]
    print(f"  stream {s}: {len(mine)} launches, kernel mean {sum(durs) / len(durs):.1f} us (min {min(durs):.1f}, max {max(durs):.1f}); "
          f"gap to the stream's next step kernel mean {sum(gaps) / max(len(gaps), 1):.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f})")
others = [r for r in rows if "step_kernel" not in r[2] and t0 <= r[0] <= t1]
by = {}
for b, e, name, _ in others:
    k = name.split("(")[0][-40:]
    by.setdefault(k, []).append((e - b) / 1e3)
for k, v in by.items():
    print(f"  other kernels in the window: {k}: {len(v)} launches, mean {sum(v) / len(v):.1f} us")
