"""Dev check of the XCD-local persistent stages (hd_xcd.hpp) on an MI355X: bits against the per-GEMM launches of the same
program (hd_set_option "xcd"), the placement-independent hand-off form, reproducibility, and the step time of both."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifidiff_amd import _lib, sampling, schedulers, synth  # noqa: E402
from hifidiff_amd.refiner import FacialRefiner  # noqa: E402

torch.set_grad_enabled(False)
L = _lib.lib()


def opt(m, key, v):
    _lib.check(L.hd_set_option(m.engine.ctx, key.encode(), int(v)), m.engine.ctx)


def read(m, name):
    L.hd_debug_read.restype = ctypes.c_int64
    n = L.hd_debug_read(m.engine.ctx, name.encode(), None, 0)
    buf = np.empty(n, dtype=np.float32)
    _lib.check(L.hd_debug_read(m.engine.ctx, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), n), m.engine.ctx)
    return buf


def main():
    W = synth.refiner_state_dict(16)
    m = FacialRefiner(16); m.load_state_dict(W); m.to("cuda:0")
    batches = [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["64", "5", "2", "13"])]
    for B in batches:
        x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
        opt(m, "xcd", 1)
        e1 = m(x, 500, crf, crl).sample.clone()
        print(f"B={B}: xcd effective {L.hd_get_option(m.engine.ctx, b'xcd')}, stages {L.hd_get_option(m.engine.ctx, b'xcd_stages')}, ops {L.hd_num_ops(m.engine.ctx, 0)}", flush=True)
        bufs1 = {k: read(m, k) for k in ("X2", "X3", "G2", "G3")}
        e1b = m(x, 500, crf, crl).sample.clone()
        opt(m, "xcd_force_global", 1)
        e1g = m(x, 500, crf, crl).sample.clone()
        opt(m, "xcd_force_global", 0)
        opt(m, "xcd", 0)
        e0 = m(x, 500, crf, crl).sample.clone()
        bufs0 = {k: read(m, k) for k in ("X2", "X3", "G2", "G3")}
        rel = float((e1 - e0).norm() / e0.norm())
        print(f"B={B}: eps equal bits xcd vs launches: {torch.equal(e1, e0)} (rel {rel:.2e}); reproducible {torch.equal(e1, e1b)}; "
              f"global hand-off form equal {torch.equal(e1g, e0)}; finite {bool(torch.isfinite(e1).all())}", flush=True)
        for k in bufs1:
            d = np.abs(bufs1[k] - bufs0[k]).max()
            print(f"      {k}: max abs diff {d:.3e}")
        if not torch.equal(e1, e0):
            # bisect by phase: stop every stage after n phases and compare the hand-off buffers of level 2
            for n in range(1, 11):
                opt(m, "xcd", 1); opt(m, "xcd_phase_limit", n)
                m(x, 500, crf, crl)
                a = {k: read(m, k) for k in ("G2", "X2")}
                print(f"      phase_limit {n}: G2 sum {a['G2'].astype(np.float64).sum():.6f} X2 sum {a['X2'].astype(np.float64).sum():.6f}")
            opt(m, "xcd_phase_limit", 0)
    # step time, both forms
    B = batches[0]
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:100]
    for on in (1, 0, 1):
        opt(m, "xcd", on)
        out = sampling.sample(m, x, crf, crl, sch, seed=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out2 = sampling.sample(m, x, crf, crl, sch, seed=3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B} xcd={on}: {dt / 100 * 1e3:.4f} ms per step (100 steps, wall), reproducible {torch.equal(out, out2)}", flush=True)
        if on == 1:
            keep = out.clone()
        else:
            print(f"      100-step latents equal bits xcd vs launches: {torch.equal(keep, out)}")


if __name__ == "__main__":
    main()
