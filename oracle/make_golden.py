#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Runs only in the build container (the reference does not exist on the GPU box and never enters
this repo).  The reference's `models/denoiser/model.py:4` imports `diffusers.ConfigMixin`, which is
not installed; it is used purely as an empty attribute bag (model.py:39-41,144-146), so a stub
module providing `class ConfigMixin: pass` is registered before the import.

Weights and inputs come from hifidiff_amd.synth (regenerable anywhere from one seed); the fixtures
hold only the reference's OUTPUTS, plus the shapes/tags needed to regenerate the inputs.

The scheduler is NOT reference code (diffusers is absent): the end-to-end fixtures run the
reference network inside oracle.hifidiff_oracle's restated DDIM/DDPM schedulers, so they pin the
network-in-the-loop, not the scheduler arithmetic ("parity unpinned" for the scheduler).

Usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from hifidiff_amd import arch, synth          # noqa: E402
from oracle import hifidiff_oracle as O       # noqa: E402


def import_reference(ref):
    stub = types.ModuleType("diffusers")
    stub.ConfigMixin = type("ConfigMixin", (), {})
    sys.modules["diffusers"] = stub
    sys.path.insert(0, ref)
    from models.refiner import FacialRefiner
    from models.denoiser.conditional_naf import ConditionalNAFBlock
    from models.denoiser.model import SinusoidalPosEmb
    from models.fpg.hca import HybridCrossAttention
    return FacialRefiner, ConditionalNAFBlock, SinusoidalPosEmb, HybridCrossAttention


def sub_state(sd, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@torch.no_grad()
def gen_uncond(args):
    """Unconditional `Denoiser(16)` (models/denoiser/model.py:32-134): eps at three timesteps and a 10-step DDIM
    (pretrain_denoiser.py:76-120 loop body inside the restated scheduler).  Weights: the `denoiser.*` tensors of the
    synthetic refiner state dict that the Denoiser has (no hcas / idc_conv)."""
    import_reference(args.ref)
    from models.denoiser.model import Denoiser
    man = arch.denoiser_manifest(16, prefix="denoiser", fused=False)
    sd = {k[len("denoiser."):]: torch.from_numpy(v) if isinstance(v, np.ndarray) else v
          for k, v in synth.make_state_dict(man).items()}
    net = Denoiser(16).eval()
    net.load_state_dict(sd, strict=True)
    B = 2
    x = T(np.stack([synth.randn(f"x_T/{f}", (4, 16, 16)) for f in range(B)]))
    out = {}
    for i, t in enumerate([(980, 980), (500, 20), (0, 0)]):
        tt = torch.tensor(t)
        out[f"t{i}"] = tt.numpy()
        out[f"eps{i}"] = net(x, tt).sample.numpy()
    out["eps_scalar_t"] = net(x, 321).sample.numpy()           # python scalar timestep (model.py:113-114)
    sch = O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
    sch.set_timesteps(10)
    lat = x.clone()
    for t in sch.timesteps:
        eps = net(lat, torch.full((B,), int(t))).sample
        lat = sch.step(eps, int(t), lat).prev_sample
    out["ddim10"] = lat.numpy()
    np.savez_compressed(os.path.join(args.out, "denoiser_uncond_L16.npz"), **out)
    print("wrote denoiser_uncond_L16.npz", {k: v.shape for k, v in out.items()})


@torch.no_grad()
def gen_l32_loop(args):
    """BASELINE configs[3] in miniature: FacialRefiner(32) (32->256 px), the first 20 steps of the 250-step DDIM schedule
    (eta 0, clip 3.0), one face: the reference network inside the restated scheduler."""
    FacialRefiner = import_reference(args.ref)[0]
    net = FacialRefiner(32).eval()
    net.load_state_dict(synth.refiner_state_dict(32), strict=True)
    x, crl, crf = synth.sample_inputs(1, 32)
    sch = O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
    sch.set_timesteps(250)
    lat = x.clone()
    for t in sch.timesteps[:20]:
        eps = net(lat, torch.full((1,), int(t)), crf, crl).sample
        lat = sch.step(eps, int(t), lat).prev_sample
    np.savez_compressed(os.path.join(args.out, "ddim250_first20_L32.npz"), final=lat.numpy())
    print("wrote ddim250_first20_L32.npz", lat.shape, float(lat.abs().max()))


@torch.no_grad()
def gen_cr(args):
    """CoarseRestoration (models/cr/model.py:33-88) on two synthetic 128x128 faces: output, the nine STN thetas and
    the stage outputs.  The synthetic fc_loc.2 weights are scaled down so that theta stays near the identity the
    reference initialises it to (stn.py:38-41) instead of sampling mostly outside the image."""
    import_reference(args.ref)
    from models.cr.model import CoarseRestoration
    sd = synth.cr_state_dict()
    net = CoarseRestoration().eval()
    net.load_state_dict(sd, strict=True)
    x = T(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(2)]))
    out = {"out": net(x).numpy()}
    # stage taps through forward hooks: thetas and stage outputs
    feats = {}
    h = net.intro(x)
    skips = []
    for i, enc in enumerate(net.encoders):
        h = enc(h); skips.append(h); feats[f"encoders.{i}"] = h
    h = net.middle_blocks(h); feats["middle_blocks"] = h
    for i, (dec, sk) in enumerate(zip(net.decoders, skips[::-1])):
        h = dec(h + sk); feats[f"decoders.{i}"] = h
    assert torch.equal(net.outro(h), torch.from_numpy(out["out"]))
    for k in ("encoders.3", "middle_blocks"):                      # small maps only: the fixture stays < 1 MB
        out["feat." + k] = feats[k].numpy()
    np.savez_compressed(os.path.join(args.out, "coarse_restoration.npz"), **out)
    print("wrote coarse_restoration.npz", {k: v.shape for k, v in out.items()})


@torch.no_grad()
def gen_cr_wild(args):
    """CoarseRestoration with strong STN warps (synth.cr_state_dict(wild=True)): scales 0.6-1.4 and translations up to
    0.4, so the affine grids of models/cr/stn.py:43-52 sample partly outside the image (zero padding)."""
    import_reference(args.ref)
    from models.cr.model import CoarseRestoration
    net = CoarseRestoration().eval()
    net.load_state_dict(synth.cr_state_dict(wild=True), strict=True)
    x = T(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(2)]))
    thetas = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: thetas.append(out.detach().reshape(-1, 6).clone()))
             for n, m in net.named_modules() if n.endswith("stn.fc_loc")]
    out = {"out": net(x).numpy()}
    for h in hooks:
        h.remove()
    out["thetas"] = torch.stack(thetas).numpy()                       # [9 STNs, 2 faces, 6]
    th = torch.stack(thetas)
    # fraction of grid points outside [-1, 1] for the first STN (128 x 128): must be substantial for the fixture to bite
    g = torch.nn.functional.affine_grid(th[0].reshape(-1, 2, 3), (2, 1, 128, 128), align_corners=False)
    out["outside_frac_stn0"] = np.float32(((g.abs() > 1).any(-1)).float().mean())
    np.savez_compressed(os.path.join(args.out, "coarse_restoration_wild.npz"), **out)
    print("wrote coarse_restoration_wild.npz", {k: getattr(v, "shape", v) for k, v in out.items()}, "outside", out["outside_frac_stn0"])


@torch.no_grad()
def gen_ddpm_slices(args):
    """Two 20-step slices of the 1000-step DDPM schedule (clip 3.0, fixed_small variance, committed-seed noise), B=2, the
    reference network inside the restated scheduler: the TAIL t = 19..0 (ends with the no-noise step t = 0) and a
    MID-trajectory slice t = 519..500.  Start latents are synthetic ("x_tail/f" x 0.7, "x_mid/f")."""
    FacialRefiner = import_reference(args.ref)[0]
    net = FacialRefiner(16).eval()
    net.load_state_dict(synth.refiner_state_dict(16), strict=True)
    _, crl2, crf2 = synth.sample_inputs(2, 16)
    out = {}
    for name, first, scale in (("tail", 980, 0.7), ("mid", 480, 1.0)):
        sch = O.DDPMScheduler(clip_sample=True, clip_sample_range=3.0)
        ts = sch.timesteps[first:first + 20]
        lat = T(np.stack([np.float32(scale) * synth.randn(f"x_{name}/{f}", (4, 16, 16)) for f in range(2)]))
        for i, t in enumerate(ts):
            eps = net(lat, torch.full((2,), t), crf2, crl2).sample
            z = T(np.stack([synth.ddpm_noise(first + i, b, 16) for b in range(2)]))
            lat = sch.step(eps, t, lat, noise=z).prev_sample
        out[name] = lat.numpy()
        out[name + "_t"] = np.array(ts)
        print(name, ts[0], ts[-1], float(lat.abs().max()))
    np.savez_compressed(os.path.join(args.out, "ddpm_slices_L16.npz"), **out)


@torch.no_grad()
def gen_full_trajectories(args, which):
    """The full loops of BASELINE configs[1] / configs[3] in miniature, reference network inside the restated schedulers:
      ddpm1000: FacialRefiner(16), B = 2, all 1000 DDPM steps (clip 3.0, fixed_small variance, committed-seed noise
                synth.ddpm_noise(step, face, 16)); the latent after every 100 steps and the final one;
      ddim250:  FacialRefiner(32), B = 1, all 250 DDIM steps (eta 0, clip 3.0); the latent after every 50 steps and the final one.
    The conditioning (fpg, idc) is evaluated once and passed to `net.denoiser(...)` (models/refiner.py:33-38 computes the same
    tensors again in every step: identical values, 2.5x the CPU time)."""
    FacialRefiner = import_reference(args.ref)[0]
    L, B, kind, n, every = (16, 2, "ddpm", 1000, 100) if which == "ddpm1000" else (32, 1, "ddim", 250, 50)
    net = FacialRefiner(L).eval()
    net.load_state_dict(synth.refiner_state_dict(L), strict=True)
    x, crl, crf = synth.sample_inputs(B, L)
    pri, emb = net.fpg(crl), net.idc(crf)
    if kind == "ddpm":
        sch = O.DDPMScheduler(clip_sample=True, clip_sample_range=3.0)
    else:
        sch = O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
        sch.set_timesteps(n)
    ts = list(sch.timesteps)
    assert len(ts) == n
    lat = x.clone()
    out = {}
    t0 = time.time()
    for i, t in enumerate(ts):
        eps = net.denoiser(lat, torch.full((B,), int(t)), pri, emb).sample
        if kind == "ddpm":
            z = T(np.stack([synth.ddpm_noise(i, b, L) for b in range(B)]))
            lat = sch.step(eps, int(t), lat, noise=z).prev_sample
        else:
            lat = sch.step(eps, int(t), lat).prev_sample
        if (i + 1) % every == 0:
            out[f"step{i + 1}"] = lat.numpy().copy()
            print(which, i + 1, f"{time.time() - t0:.0f}s", float(lat.abs().max()), float(lat.std()), flush=True)
    out["final"] = lat.numpy()
    name = "ddpm1000_L16.npz" if which == "ddpm1000" else "ddim250_L32.npz"
    np.savez_compressed(os.path.join(args.out, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-l32", action="store_true")
    ap.add_argument("--only", default="", help="'uncond': only the unconditional Denoiser fixture")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    if args.only == "uncond":
        return gen_uncond(args)
    if args.only == "cr":
        return gen_cr(args)
    if args.only == "l32loop":
        return gen_l32_loop(args)
    if args.only == "crwild":
        return gen_cr_wild(args)
    if args.only == "ddpmslices":
        return gen_ddpm_slices(args)
    if args.only in ("ddpm1000", "ddim250"):
        return gen_full_trajectories(args, args.only)
    FacialRefiner, CondBlock, PosEmb, HCA = import_reference(args.ref)

    t0 = time.time()
    sd = synth.refiner_state_dict(16)
    net = FacialRefiner(16).eval()
    net.load_state_dict(sd, strict=True)
    print(f"reference model ready in {time.time() - t0:.1f}s")

    # ---- 1. one ConditionalNAFBlock per level (conditional_naf.py:108-136) ----
    out = {}
    temb = T(synth.randn("blk_temb", (1, 512)))
    names = ["encoders.0.0", "encoders.1.0", "encoders.2.0", "encoders.3.0", "middle_blks.0"]
    for l, (c, h) in enumerate(arch.naf_levels(16)):
        blk = CondBlock(c, 512).eval()
        blk.load_state_dict(sub_state(sd, "denoiser." + names[l]), strict=True)
        x = T(synth.randn(f"blk_in/{l}", (1, c, h, h)))
        y, _ = blk([x, temb])
        out[f"out{l}"] = y.numpy()
    np.savez_compressed(os.path.join(args.out, "cond_naf_blocks.npz"), **out)

    # ---- 2. HybridCrossAttention at every resolution (hca.py:25-48) ----
    out = {}
    for i, (c, h) in enumerate(arch.naf_levels(16)[::-1]):
        m = HCA(c).eval()
        m.load_state_dict(sub_state(sd, f"denoiser.hcas.{i}"), strict=True)
        f_g = T(synth.randn(f"hca_fg/{i}", (1, c, h, h)))
        f_d = T(synth.randn(f"hca_fd/{i}", (1, c, h, h)))
        out[f"out{i}"] = m(f_g, f_d).numpy()
        out[f"wc{i}"] = m.channel_cross_attention(f_g).numpy()
        out[f"ws{i}"] = m.spatial_cross_attention(f_g).numpy()
    np.savez_compressed(os.path.join(args.out, "hca.npz"), **out)

    # ---- 3. sinusoidal embedding + time_mlp (model.py:22-29,152-157) ----
    ts = torch.tensor([0.0, 1.0, 500.0, 999.0])
    np.savez_compressed(os.path.join(args.out, "time_embedding.npz"),
                        t=ts.numpy(), posemb=PosEmb(128)(ts).numpy(),
                        temb=net.denoiser.time_mlp(ts).numpy())

    # ---- 4./5. FPG priors and ResNet-50 embedding, B=1 ----
    x1, crl1, crf1 = synth.sample_inputs(1, 16)
    pri = net.fpg(crl1)
    np.savez_compressed(os.path.join(args.out, "fpg_priors.npz"),
                        **{f"prior{i}": p.numpy() for i, p in enumerate(pri)})
    np.savez_compressed(os.path.join(args.out, "idc_embedding.npz"), emb=net.idc(crf1).numpy())

    # ---- 6. full refiner eps, B=2, a few timesteps (refiner.py:32-38) ----
    x2, crl2, crf2 = synth.sample_inputs(2, 16)
    out = {}
    for t in (980, 500, 0):
        out[f"eps_t{t}"] = net(x2, torch.full((2,), t), crf2, crl2).sample.numpy()
    # mixed per-face timesteps and the scalar form accepted by model.py:218-229
    out["eps_tmixed"] = net(x2, torch.tensor([37, 861]), crf2, crl2).sample.numpy()
    pri2 = net.fpg(crl2)
    emb2 = net.idc(crf2)
    out["eps_scalar_t250"] = net.denoiser(x2, 250, pri2, emb2).sample.numpy()
    np.savez_compressed(os.path.join(args.out, "refiner_eps_L16.npz"), **out)

    # ---- 7. 50-step DDIM, clip 3.0, as written (test_refiner.py:85-91,166-171), B=2 ----
    t0 = time.time()
    sch = O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
    sch.set_timesteps(50)
    lat = x2.clone()
    for t in sch.timesteps:
        eps = net(lat, torch.full((2,), t), crf2, crl2).sample
        lat = sch.step(eps, t, lat, eta=0.0).prev_sample
    np.savez_compressed(os.path.join(args.out, "ddim50_L16.npz"), final=lat.numpy())
    print(f"ddim50 {time.time() - t0:.1f}s  |x|max={lat.abs().max():.3f}")

    # ---- 8. first 20 steps of the 1000-step DDPM (clip 3.0) with committed-seed noise, B=2 ----
    t0 = time.time()
    sch = O.DDPMScheduler(clip_sample=True, clip_sample_range=3.0)
    lat = x2.clone()
    for i, t in enumerate(sch.timesteps[:20]):
        eps = net(lat, torch.full((2,), t), crf2, crl2).sample
        z = T(np.stack([synth.ddpm_noise(i, b, 16) for b in range(2)]))
        lat = sch.step(eps, t, lat, noise=z).prev_sample
    np.savez_compressed(os.path.join(args.out, "ddpm20_L16.npz"), final=lat.numpy())
    print(f"ddpm20 {time.time() - t0:.1f}s  |x|max={lat.abs().max():.3f}")

    # ---- 9. latent 32 (FacialRefiner(32): idc_conv 2048->8192, mid at 2x2), B=1 ----
    if not args.skip_l32:
        del net
        sd32 = synth.refiner_state_dict(32)
        net32 = FacialRefiner(32).eval()
        net32.load_state_dict(sd32, strict=True)
        x, crl, crf = synth.sample_inputs(1, 32)
        e = net32(x, torch.full((1,), 500), crf, crl).sample
        np.savez_compressed(os.path.join(args.out, "refiner_eps_L32.npz"), eps_t500=e.numpy())
    print("done")


if __name__ == "__main__":
    main()
