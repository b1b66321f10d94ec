"""Dev check of the face-cluster stages of levels 0 / 1 (hd_face.hpp) on an MI355X: eps against the per-block launches of the
same program (hd_set_option "face"; not bit-identical: two-pass LayerNorm statistics instead of merged partials), against the
bf16-emulating oracle, reproducibility, and the step time of both forms."""
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
from oracle import hifidiff_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
L = _lib.lib()
L.hd_debug_read.restype = ctypes.c_int64


def opt(m, key, v):
    _lib.check(L.hd_set_option(m.engine.ctx, key.encode(), int(v)), m.engine.ctx)


def read(m, name):
    n = L.hd_debug_read(m.engine.ctx, name.encode(), None, 0)
    buf = np.empty(n, dtype=np.float32)
    _lib.check(L.hd_debug_read(m.engine.ctx, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), n), m.engine.ctx)
    return buf


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    W = synth.refiner_state_dict(16)
    m = FacialRefiner(16); m.load_state_dict(W); m.to("cuda:0")
    batches = [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["64", "5", "2"])]
    for B in batches:
        x, crl, crf = synth.sample_inputs(B, 16)
        xd, cld, cfd = x.cuda(), crl.cuda(), crf.cuda()
        names = None
        opt(m, "face", 1)
        e1 = m(xd, 500, cfd, cld).sample.clone()
        names = [L.hd_debug_op_name(m.engine.ctx, 0, i).decode() for i in range(L.hd_num_ops(m.engine.ctx, 0))]
        print(f"B={B}: ops {len(names)}, face stages {L.hd_get_option(m.engine.ctx, b'face_stages')}", flush=True)
        e1b = m(xd, 500, cfd, cld).sample.clone()
        # the first stage alone: X0 after encoders.0.1
        i0 = names.index("denoiser.encoders.0.1.conv5")
        L.hd_debug_limit_ops(m.engine.ctx, 0, i0 + 1)
        m(xd, 500, cfd, cld); xa = torch.from_numpy(read(m, "X0")[:B * 256 * 128].copy())
        opt(m, "face", 0)
        m(xd, 500, cfd, cld); xb = torch.from_numpy(read(m, "X0")[:B * 256 * 128].copy())
        L.hd_debug_limit_ops(m.engine.ctx, 0, -1)
        e0 = m(xd, 500, cfd, cld).sample.clone()
        print(f"B={B}: X0 after the first stage: rel {rel(xa, xb):.3e} max abs {float((xa - xb).abs().max()):.3e}; eps face vs launches rel {rel(e1.cpu(), e0.cpu()):.3e}; "
              f"reproducible {torch.equal(e1, e1b)}; finite {bool(torch.isfinite(e1).all())}", flush=True)
        if B <= 16:
            cond = O.Conditioning(W, crl, crf, prec=O.BF16)
            ref = O.fused_denoiser(W, x, 500, cond=cond, prec=O.BF16)
            print(f"      vs emulating oracle: face {rel(e1.cpu(), ref):.3e}, launches {rel(e0.cpu(), ref):.3e}", flush=True)
    B = batches[0]
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:100]
    for on in (1, 0, 1):
        opt(m, "face", on)
        out = sampling.sample(m, x, crf, crl, sch, seed=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out2 = sampling.sample(m, x, crf, crl, sch, seed=3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B} face={on}: {dt / 100 * 1e3:.4f} ms per step (100 steps, wall), reproducible {torch.equal(out, out2)}", flush=True)


if __name__ == "__main__":
    main()
