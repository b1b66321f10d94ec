#!/usr/bin/env python3
"""Per-diffusion-step HBM traffic from rocprofv3 --pmc runs of bench.py.
usage: pmc_traffic.py <counter_collection_A.csv> <stepsA> <counter_collection_B.csv> <stepsB> <COUNTER>
Two runs with different numbers of diffusion steps: (sum_B - sum_A) / (stepsB - stepsA) isolates the loop from
the prologue/packing.  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B (MI355X_MICROARCH.md §HBM);
FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 (same section), corrected here."""
import csv
import sys


def total(path, counter):
    s = 0.0
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            s += float(r["Counter_Value"])
    return s


a, na, b, nb, counter = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
per_step = (total(b, counter) - total(a, counter)) / (nb - na)
raw_bytes = per_step * 1024.0
corr = 2.0 if counter == "FETCH_SIZE" else 1.0
print(f"{counter}: {per_step:.1f} units/step -> raw {raw_bytes/1e6:.1f} MB/step, corrected x{corr:g} = {raw_bytes*corr/1e6:.1f} MB/step")
