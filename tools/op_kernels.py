#!/usr/bin/env python3
"""Which kernel runs each launch of one diffusion step: rocprofv3 --kernel-trace CSV of bench.py + the library's op list
(HD_DUMP_OPS).  usage: op_kernels.py <kernel_trace.csv> <ops.txt> [filter substring]"""
import csv
import sys

ops = [l.strip() for l in open(sys.argv[2]) if l.strip()]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the step graph repeats: find the first index where `len(ops)` consecutive kernels END with the ending conv, twice in a row
n = len(ops)
start = next(i for i in range(len(rows) - 2 * n) if "ending_conv" in names[i + n - 1] and "ending_conv" in names[i + 2 * n - 1] and (i == 0 or "ending_conv" in names[i - 1]))
for k in range(n):
    r = rows[start + n + k]                     # second replay: warm
    if flt in ops[k]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        print(f"{ops[k]:42s} {d:7.2f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')} wg {r.get('Workgroup_Size_X', '?')}  {names[start + n + k][:150]}")
