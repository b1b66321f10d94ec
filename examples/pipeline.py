#!/usr/bin/env python3
"""End-to-end latent-space pipeline of `ddim_sample` (test_refiner.py:58-95) on an MI355X, with synthetic weights:

    cr_face   = CoarseRestoration()(ln_face)                 # hifidiff_amd.cr          (test_refiner.py:77)
    cr_latent = vae.encode(bicubic(cr_face)).latent_dist.sample() * 0.18215   # hifidiff_amd.vae (test_refiner.py:78-83)
    latent    = 50-step DDIM with FacialRefiner              # hifidiff_amd.refiner + sampling (test_refiner.py:85-91)
    images    = vae.decode(latent / 0.18215).sample          # hifidiff_amd.vae         (test_refiner.py:93)

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
from hifidiff_amd.vae import AutoencoderKL                               # noqa: E402


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
    vae = AutoencoderKL()
    vae.load_state_dict(synth.vae_state_dict())          # real use: AutoencoderKL.from_pretrained(<local sd-2-1-base dir>, subfolder="vae")
    vae.to(dev)
    model = FacialRefiner(latent_res=16)
    model.load_state_dict(synth.refiner_state_dict(16))  # real use: safetensors.torch.load_file(refiner_ckpt)
    model.to(dev)
    sch = schedulers.DDIMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="epsilon",
                                   clip_sample_range=3.0)

    B = a.batch
    ln_face = torch.from_numpy(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(B)])).to(dev)
    latent = torch.randn(B, 4, 16, 16, device=dev)

    torch.cuda.synchronize(); t0 = time.time()
    cr_face = cr(ln_face)
    torch.cuda.synchronize(); t1 = time.time()
    cr_latent = vae.encode_scaled(cr_face, 128, seed=7)                # bicubic (identity at 128) + encode + sample + x 0.18215
    torch.cuda.synchronize(); t2 = time.time()
    sch.set_timesteps(a.steps)
    out = sampling.sample(model, latent, cr_face, cr_latent, sch)      # conditioning once + graph-replayed loop
    torch.cuda.synchronize(); t3 = time.time()
    images = vae.decode(out / 0.18215).sample                          # the reference's call form; decode_scaled(out) is the fused one
    torch.cuda.synchronize(); t4 = time.time()
    print(f"batch {B}: coarse restoration {1e3 * (t1 - t0):.1f} ms, VAE encode {1e3 * (t2 - t1):.1f} ms, {a.steps}-step DDIM "
          f"{1e3 * (t3 - t2):.1f} ms, VAE decode {1e3 * (t4 - t3):.1f} ms; latent range [{float(out.min()):.2f}, {float(out.max()):.2f}], "
          f"images {tuple(images.shape)} finite {bool(torch.isfinite(images).all())}")


if __name__ == "__main__":
    main()
