#!/usr/bin/env python3
"""Where the conditioning prologue's time goes (FPG, ResNet-50 IDC, HCA gates, idc_conv: once per reverse pass, inside bench.py's timed
region): the prologue program is run up to launch i (hd_debug_limit_ops on program 1) and timed with events; the differences are the launches.
usage (GPU): python tools/prologue_times.py [batch] [latent] [out]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifidiff_amd import _lib, synth                           # noqa: E402
from hifidiff_amd.refiner import FacialRefiner                 # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    latent = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "prologue_times.txt")
    torch.set_grad_enabled(False)
    m = FacialRefiner(latent)
    m.load_state_dict(synth.refiner_state_dict(latent))
    m.to("cuda")
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, latent)]
    e, L = m.engine, _lib.lib()
    e.prepare(crl, cr_face=crf)
    n = L.hd_num_ops(e.ctx, 1)
    names = [L.hd_debug_op_name(e.ctx, 1, i).decode() for i in range(n)]

    def timed(limit, reps=5):
        L.hd_debug_limit_ops(e.ctx, 1, limit)
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            e.prepare(crl, cr_face=crf)
            t1.record()
            torch.cuda.synchronize()
            best = min(best, t0.elapsed_time(t1))
        return best

    total = timed(-1)
    cum = [timed(i + 1, 3) for i in range(n)]
    L.hd_debug_limit_ops(e.ctx, 1, -1)
    lines = [f"conditioning prologue, batch {B}, latent {latent}: {n} launches, {total:.3f} ms (events around hd_prepare, best of 5)"]
    groups = {}
    prev = 0.0
    for name, c in zip(names, cum):
        d = max(c - prev, 0.0)
        prev = max(prev, c)
        key = name.split(".")[0] if not name.startswith("idc.layer") else ".".join(name.split(".")[:2])
        if name.startswith("fpg."):
            key = ".".join(name.split(".")[:3]) if name.split(".")[1] in ("encoders",) else ".".join(name.split(".")[:2])
        g = groups.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += d
    for k, (cnt, t) in groups.items():
        lines.append(f"  {k:32s} {cnt:4d} launches {t * 1e3:9.1f} us")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
