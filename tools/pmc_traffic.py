#!/usr/bin/env python3
"""Per-diffusion-step HBM traffic from rocprofv3 --pmc runs of bench.py.
usage: pmc_traffic.py <counter_collection_A.csv> <stepsA> <counter_collection_B.csv> <stepsB> <COUNTER> [out.json]
Two runs with different numbers of diffusion steps: (sum_B - sum_A) / (stepsB - stepsA) isolates the loop from
the prologue/packing.  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B (MI355X_MICROARCH.md §HBM);
FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 (same section), corrected here.
With out.json the counter's entry is merged into that file (the file bench.py reads `traffic` from)."""
import csv
import json
import os
import sys


def total(path, counter):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            s += float(r["Counter_Value"])
            n += 1
    return s, n


a, na, b, nb, counter = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
(sa, da), (sb, db) = total(a, counter), total(b, counter)
per_step = (sb - sa) / (nb - na)
raw_bytes = per_step * 1024.0
corr = 2.0 if counter == "FETCH_SIZE" else 1.0
print(f"{counter}: {per_step:.1f} units/step -> raw {raw_bytes/1e6:.1f} MB/step, corrected x{corr:g} = {raw_bytes*corr/1e6:.1f} MB/step")
if len(sys.argv) > 6:
    path = sys.argv[6]
    j = json.load(open(path)) if os.path.exists(path) else {"counters": {}}
    j["counters"][counter] = {"units_per_step": per_step, "raw_bytes_per_step": raw_bytes, "correction": corr,
                              "bytes_per_step": raw_bytes * corr, "dispatches_%d" % na: da, "dispatches_%d" % nb: db}
    j["hbm_bytes_per_step"] = sum(c["bytes_per_step"] for c in j["counters"].values())
    j["algorithmic_bytes_per_step"] = 722.66e6
    j["source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) of "
                   "`bench.py --steps 1 --warmup 0 --diffusion-steps {%d,%d}`; per-step = difference/%d; FETCH_SIZE x2 "
                   "(gfx950 under-count of wide coalesced reads, MI355X_MICROARCH.md HBM section); batch 64, latent 16"
                   % (na, nb, nb - na))
    json.dump(j, open(path, "w"), indent=1)
