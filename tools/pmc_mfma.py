#!/usr/bin/env python3
"""MFMA utilisation of the bench command from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE;
optionally SQ_INSTS_VALU_MFMA_MOPS_BF16), per kernel family and for the whole diffusion step.
usage: pmc_mfma.py <counter_collection.csv> <ops.txt> <latent> <kind> [out.json]

MFMA busy fraction of a dispatch = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the counter sums the cycles in
which a SIMD's matrix pipe is busy over all 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md cycle constants);
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (same guide, DVFS section).  Only the dispatches of complete diffusion
steps are counted (the op list gives the launches per step; the conditioning prologue and weight packing are left out)."""
import collections
import csv
import json
import re
import sys


def family(k):
    m = re.search(r"xcd(2?)_stage_kernel<(\d+), *(\d+)>", k)
    if m:
        return f"xcd{m.group(1)}_stage<{m.group(2)},{m.group(3)}> (levels 2/3, persistent)"
    m = re.search(r"naf_face_stage_kernel<(\d+), *(\d+)>", k)
    if m:
        return f"naf_face_stage<{m.group(1)},{m.group(2)}> (levels 0/1, persistent)"
    m = re.search(r"gemm_wide_kernel<(true|false), *(?:hd::)?(Ep\w+), *(true|false), *(\d+), *(\d+)>", k)
    if m:
        form = {"0": "128-row", "1": "256-row", "2": "64-row"}.get(m.group(4), "form " + m.group(4))
        return f"gemm_wide {'LN' if m.group(1) == 'true' else 'bf16'} {m.group(2)} {form} (latent 32, role-split)"
    if "gemm_wide_kernel" in k:
        return "gemm_wide (latent 32, role-split)"
    if "gemm_deep" in k:
        ld = "LN" if "LdF32LN" in k else "bf16"
        ep = re.search(r"hd::(Ep\w+)", k)
        return f"gemm_deep{'_pair8' if 'pair8' in k else ''} {ld} {ep.group(1) if ep else ''} (many-row)"
    m = re.search(r"naf_strip_dwgate_kernel<(\d+), *(\d+)>", k)
    if m:
        return f"naf_strip_dwgate<{m.group(1)},{m.group(2)}> (latent 32, levels 0/1)"
    if "naf_chain_kernel" in k:
        return "naf_chain (levels 0/1)"
    if "hca_ending_conv_kernel" in k:
        return "hca_conv 3x3 + ending conv (one launch)"
    if "hca_conv_kernel" in k:
        return "hca_conv 3x3"
    if "intro_conv" in k or "ending_conv" in k:
        return "intro / ending conv (fp32 FMA)"
    if "gemm_skinny_kernel" in k or "gemm_kernel" in k:
        ld = "LN" if "LdF32LN" in k else ("conv-gather" if "LdConv" in k else "bf16")
        ep = re.search(r"hd::(Ep\w+)", k)
        return f"gemm {ld} {ep.group(1) if ep else ''}"
    return "other"


def main():
    path, ops_path, latent, kind = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    n_ops = len([l for l in open(ops_path) if l.strip()])
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda d: d["t0"])
    # a diffusion step ends with the ending conv (the intro conv may be folded into the first stage)
    ends = [i for i, d in enumerate(rows) if "ending_conv_kernel" in d["name"]]
    starts = [j - n_ops + 1 for i, j in zip(ends, ends[1:]) if j - i == n_ops]
    if not starts:
        sys.exit("no complete diffusion step found (%d launches per step, %d dispatches)" % (n_ops, len(rows)))
    sel = [rows[s + k] for s in starts for k in range(n_ops)]
    fam = collections.OrderedDict()
    tot = {"busy": 0.0, "avail": 0.0, "us": 0.0, "mops": 0.0, "sqbusy": 0.0}
    for d in sel:
        f = fam.setdefault(family(d["name"]), {"n": 0, "busy": 0.0, "avail": 0.0, "us": 0.0, "mops": 0.0})
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        avail = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
        us = (d["t1"] - d["t0"]) / 1e3
        for t in (f, tot):
            t["busy"] += busy; t["avail"] += avail; t["us"] += us; t["mops"] += d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        f["n"] += 1
        tot["sqbusy"] += d.get("SQ_BUSY_CYCLES", 0.0)
    ns = len(starts)
    print(f"latent {latent} {kind}: {ns} diffusion steps x {n_ops} launches under --pmc (kernel time {tot['us'] / ns:.1f} us per step while profiled)")
    print(f"{'kernel family':44s} {'n/step':>6s} {'us/step':>9s} {'MFMA busy':>10s} {'share of MFMA cycles':>21s}")
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
        print(f"{k:44s} {f['n'] / ns:6.0f} {f['us'] / ns:9.1f} {100.0 * f['busy'] / max(f['avail'], 1.0):9.2f}% {100.0 * f['busy'] / max(tot['busy'], 1.0):20.1f}%")
    frac = tot["busy"] / max(tot["avail"], 1.0)
    print(f"{'whole step':44s} {n_ops:6d} {tot['us'] / ns:9.1f} {100.0 * frac:9.2f}%")
    print(f"SQ_VALU_MFMA_BUSY_CYCLES per step {tot['busy'] / ns:.4g} = {tot['busy'] / ns / 32:.4g} v_mfma_f32_32x32x16_bf16-equivalents; "
          f"SQ_INSTS_VALU_MFMA_MOPS_BF16 per step {tot['mops'] / ns:.4g}")
    if len(sys.argv) > 5:
        json.dump({"mfma_busy_frac": frac, "mfma_busy_cycles_per_step": tot["busy"] / ns, "mops_bf16_per_step": tot["mops"] / ns,
                   "steps_counted": ns, "launches_per_step": n_ops, "latent": latent, "kind": kind,
                   "per_family": {k: {"launches_per_step": f["n"] / ns, "us_per_step": f["us"] / ns, "mfma_busy_frac": f["busy"] / max(f["avail"], 1.0)} for k, f in fam.items()},
                   "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace of "
                             "`bench.py --steps 1 --warmup 0 --diffusion-steps 20`; busy / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) over the step's launches"},
                  open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
