#!/usr/bin/env python3
"""End-to-end latent-space pipeline of `ddim_sample` (test_refiner.py:58-95) on an MI355X, with synthetic weights:

    cr_face   = CoarseRestoration()(ln_face)                 # hifidiff_amd.cr          (test_refiner.py:77)
    cr_latent = VAE.encode(cr_face) * 0.18215                # NOT part of this library: synthetic here (§8 f2)
    latent    = 50-step DDIM with FacialRefiner              # hifidiff_amd.refiner + sampling (test_refiner.py:85-91)
    image     = VAE.decode(latent / 0.18215)                 # NOT part of this library

    python examples/pipeline.py [--batch 8] [--steps 50]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hifidiff_amd import sampling, schedulers, synth                      # noqa: E402
from hifidiff_amd.cr import CoarseRestoration                            # noqa: E402
from hifidiff_amd.refiner import FacialRefiner                           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)

    cr = CoarseRestoration()
    cr.load_state_dict(synth.cr_state_dict())            # real use: torch.load(cr_ckpt)["model_state_dict"]
    cr.to(dev)
    model = FacialRefiner(latent_res=16)
    model.load_state_dict(synth.refiner_state_dict(16))  # real use: safetensors.torch.load_file(refiner_ckpt)
    model.to(dev)
    sch = schedulers.DDIMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="epsilon",
                                   clip_sample_range=3.0)

    B = a.batch
    ln_face = torch.from_numpy(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(B)])).to(dev)
    latent = torch.randn(B, 4, 16, 16, device=dev)
    cr_latent = 0.8 * torch.randn(B, 4, 16, 16, device=dev)          # stands in for vae.encode(cr_face) * 0.18215

    torch.cuda.synchronize(); t0 = time.time()
    cr_face = cr(ln_face)
    torch.cuda.synchronize(); t1 = time.time()
    sch.set_timesteps(a.steps)
    out = sampling.sample(model, latent, cr_face, cr_latent, sch)      # conditioning once + graph-replayed loop
    torch.cuda.synchronize(); t2 = time.time()
    print(f"batch {B}: coarse restoration {1e3 * (t1 - t0):.1f} ms, {a.steps}-step DDIM {1e3 * (t2 - t1):.1f} ms, "
          f"latent range [{float(out.min()):.2f}, {float(out.max()):.2f}], finite {bool(torch.isfinite(out).all())}")


if __name__ == "__main__":
    main()
