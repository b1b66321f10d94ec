#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals and, given the library's op
order (bench.py with HD_DUMP_OPS=file), the average time of every launch inside one diffusion step.
usage: prof_summary.py <kernel_trace.csv> [ops.txt]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
tot = collections.defaultdict(lambda: [0, 0.0])
for r, d in zip(rows, dur):
    tot[r["Kernel_Name"]][0] += 1
    tot[r["Kernel_Name"]][1] += d
allt = sum(dur)
print(f"{len(rows)} dispatches, {allt/1e3:.2f} ms total kernel time")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{v[1]/allt*100:5.1f}%  n={v[0]:6d}  avg {v[1]/v[0]:9.2f} us  {k[:140]}")
if len(sys.argv) > 2:
    ops = [l.strip() for l in open(sys.argv[2]) if l.strip()]
    n = len(ops)
    # a diffusion step ends with the ending conv (the intro conv may be folded into the first stage)
    ends = [i for i, r in enumerate(rows) if "ending_conv_kernel" in r["Kernel_Name"]]
    starts = [j - n + 1 for i, j in zip(ends, ends[1:]) if j - i == n]
    starts = starts[1:] if len(starts) > 2 else starts           # skip the first (cold) step
    if not starts:
        sys.exit("no full steps found (ops per step %d)" % n)
    avg = [sum(dur[s + k] for s in starts) / len(starts) for k in range(n)]
    gap = []
    for s in starts:
        for k in range(n - 1):
            gap.append((int(rows[s + k + 1]["Start_Timestamp"]) - int(rows[s + k]["End_Timestamp"])) / 1e3)
    wall = sum((int(rows[s + n - 1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3 for s in starts) / len(starts)
    print(f"\n{len(starts)} steps x {n} launches: kernel time {sum(avg):.1f} us/step, wall {wall:.1f} us/step, mean gap {sum(gap)/len(gap):.2f} us")
    groups = collections.OrderedDict()
    for name, a in zip(ops, avg):
        parts = name.split(".")
        key = ".".join(parts[:3]) if parts[0] == "denoiser" else parts[0] + ("." + parts[1] if len(parts) > 1 and parts[1].isdigit() else "")
        lvl = key
        suffix = parts[-1] if parts[0] == "denoiser" else name
        groups.setdefault((lvl, suffix), []).append(a)
    lv = collections.OrderedDict()
    for (lvl, suffix), v in groups.items():
        lv.setdefault(lvl, []).append((suffix, sum(v) / len(v), len(v)))
    for lvl, items in lv.items():
        print(f"{lvl:24s} " + "  ".join(f"{s}:{a:.1f}x{c}" for s, a, c in items) + f"   | sum {sum(a*c for _, a, c in items):.1f}")
