#!/usr/bin/env python3
"""Parity figures of the round, measured on an MI355X (writes gpurun_out/parity_report.txt; copy to profiles/):
  * teacher-forced per-launch errors at batch 2 and 64 (tools/op_forced.py);
  * eps at the benchmark batch (64 faces) against the bf16-emulating oracle, t in {999, 500, 0} and per-face timesteps;
  * latent 32: eps at batch 16 against the emulating oracle;
  * loop goldens (reference network inside the restated schedulers): PSNR on the golden's own peak-to-peak and rel-L2 for the
    50-step DDIM, the 20-step DDPM slices, the full 1000-step DDPM (B = 2) and the full 250-step DDIM at latent 32 (B = 1);
  * image space: HIP-decode(HIP-latent) against oracle-decode(oracle-latent) through the synthetic-weight VAE (f2).
(Test infrastructure: uses oracle/.)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hifidiff_amd import sampling, schedulers, synth          # noqa: E402
from hifidiff_amd.refiner import FacialRefiner                 # noqa: E402
from oracle import hifidiff_oracle as O                        # noqa: E402

torch.set_grad_enabled(False)
OUT = []


def say(s):
    print(s, flush=True)
    OUT.append(s)


def rel_l2(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def psnr_pp(a, b):
    """PSNR with the golden's own peak-to-peak as the data range."""
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    rng = float(b.max() - b.min())
    return float(10.0 * torch.log10(rng ** 2 / ((a - b) ** 2).mean().clamp_min(1e-30))), rng


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    what = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else {"forced", "eps64", "l32", "loops", "full", "image"}
    P = synth.refiner_state_dict(16)
    if "forced" in what:
        import op_forced
        os.environ["HD_NO_XCD"] = "1"
        ml = FacialRefiner(16); ml.load_state_dict(P); ml.to("cuda")
        del os.environ["HD_NO_XCD"]
        for B in (2, 64):
            x, crl, crf = synth.sample_inputs(B, 16)
            rep = []
            t0 = time.time()
            w = op_forced.forced_scan(ml, P, x, crl, crf, 500.0, rep)
            say(f"teacher-forced per-launch parity, batch {B}: worst fp32 output rel-L2 {w['fp32']:.3e}, worst bf16-stored {w['bf16']:.3e} "
                f"({len(rep)} checks, {sum('<<<<<<' in r for r in rep)} over their bound, {time.time() - t0:.0f} s)")
            with open(os.path.join(ROOT, "gpurun_out", f"op_forced_B{B}.txt"), "w") as f:
                f.write("\n".join(rep) + "\n")
        del ml
    m = FacialRefiner(16); m.load_state_dict(P); m.to("cuda")
    if "eps64" in what:
        x, crl, crf = synth.sample_inputs(64, 16)
        t0 = time.time()
        cond = O.Conditioning(P, crl, crf, prec=O.BF16)
        say(f"oracle conditioning for 64 faces: {time.time() - t0:.0f} s")
        xd, cld, cfd = x.cuda(), crl.cuda(), crf.cuda()
        for t in (999, 500, 0):
            e = m(xd, torch.full((64,), t, device="cuda"), cfd, cld).sample.cpu()       # per-face tensor, all equal
            es = m(xd, t, cfd, cld).sample.cpu()                                         # scalar: the shared-row kernels
            ref = O.fused_denoiser(P, x, t, cond=cond, prec=O.BF16)
            say(f"eps batch 64 t={t}: rel-L2 vs emulating oracle {rel_l2(es, ref):.3e} (shared FiLM row), {rel_l2(e, ref):.3e} (per-face rows); worst face {max(rel_l2(es[f], ref[f]) for f in range(64)):.3e}")
        tf = (torch.arange(64) * 37 % 1000).float()
        e = m(xd, tf.cuda(), cfd, cld).sample.cpu()
        ref = O.fused_denoiser(P, x, tf, cond=cond, prec=O.BF16)
        say(f"eps batch 64 per-face timesteps: rel-L2 {rel_l2(e, ref):.3e}; worst face {max(rel_l2(e[f], ref[f]) for f in range(64)):.3e}")
    if "loops" in what or "full" in what:
        x2, crl2, crf2 = [t.cuda() for t in synth.sample_inputs(2, 16)]
    if "loops" in what:
        sch = schedulers.DDIMScheduler(clip_sample_range=3.0); sch.set_timesteps(50)
        out = sampling.sample(m, x2, crf2, crl2, sch).cpu()
        g = golden("ddim50_L16.npz")["final"]
        ps, rng = psnr_pp(out, g)
        say(f"50-step DDIM (B=2) vs reference golden: PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(out, g):.3e}, saturated at +-3: {float((np.abs(g) >= 2.999).mean()):.3f}")
        sch = schedulers.DDPMScheduler(clip_sample_range=3.0); sch.timesteps = sch.timesteps[:20]
        noise = T(np.stack([np.stack([synth.ddpm_noise(i, b, 16) for b in range(2)]) for i in range(20)]))
        out = sampling.sample(m, x2, crf2, crl2, sch, noise=noise).cpu()
        g = golden("ddpm20_L16.npz")["final"]
        ps, rng = psnr_pp(out, g)
        say(f"first 20 DDPM steps (B=2): PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(out, g):.3e}")
        gs = golden("ddpm_slices_L16.npz")
        for name, first, scale in (("tail", 980, 0.7), ("mid", 480, 1.0)):
            sch = schedulers.DDPMScheduler(clip_sample_range=3.0); sch.timesteps = sch.timesteps[first:first + 20]
            xx = T(np.stack([np.float32(scale) * synth.randn(f"x_{name}/{f}", (4, 16, 16)) for f in range(2)])).cuda()
            noise = T(np.stack([np.stack([synth.ddpm_noise(first + i, b, 16) for b in range(2)]) for i in range(20)]))
            out = sampling.sample(m, xx, crf2, crl2, sch, noise=noise).cpu()
            ps, rng = psnr_pp(out, gs[name])
            say(f"DDPM slice {name} (B=2): PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(out, gs[name]):.3e}")
    if "full" in what and os.path.exists(os.path.join(ROOT, "tests", "golden", "ddpm1000_L16.npz")):
        g = golden("ddpm1000_L16.npz")
        sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
        noise = T(np.stack([np.stack([synth.ddpm_noise(i, b, 16) for b in range(2)]) for i in range(1000)]))
        for n in (100, 500, 1000):
            s2 = schedulers.DDPMScheduler(clip_sample_range=3.0); s2.timesteps = s2.timesteps[:n]
            out = sampling.sample(m, x2, crf2, crl2, s2, noise=noise[:n]).cpu()
            ref = g[f"step{n}"]
            ps, rng = psnr_pp(out, ref)
            say(f"1000-step DDPM (B=2, committed noise) after {n} steps vs reference golden: PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(out, ref):.3e}, max abs {float((out - T(ref)).abs().max()):.3e}")
        final_hip = out
    if "l32" in what:
        P32 = synth.refiner_state_dict(32)
        m32 = FacialRefiner(32); m32.load_state_dict(P32); m32.to("cuda")
        x, crl, crf = synth.sample_inputs(16, 32)
        t0 = time.time()
        cond = O.Conditioning(P32, crl, crf, prec=O.BF16)
        ref = O.fused_denoiser(P32, x, 500, cond=cond, prec=O.BF16)
        e = m32(x.cuda(), 500, crf.cuda(), crl.cuda()).sample.cpu()
        say(f"latent 32, eps batch 16 t=500: rel-L2 vs emulating oracle {rel_l2(e, ref):.3e}; worst face {max(rel_l2(e[f], ref[f]) for f in range(16)):.3e} (oracle {time.time() - t0:.0f} s)")
        g = golden("ddim250_first20_L32.npz")["final"]
        sch = schedulers.DDIMScheduler(clip_sample_range=3.0); sch.set_timesteps(250); sch.timesteps = sch.timesteps[:20]
        x1, crl1, crf1 = synth.sample_inputs(1, 32)
        lat = sampling.sample(m32, x1.cuda(), crf1.cuda(), crl1.cuda(), sch).cpu()
        ps, rng = psnr_pp(lat, g)
        say(f"latent 32, first 20 of 250 DDIM steps (B=1): PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(lat, g):.3e}")
        if "full" in what and os.path.exists(os.path.join(ROOT, "tests", "golden", "ddim250_L32.npz")):
            gf = golden("ddim250_L32.npz")
            sch = schedulers.DDIMScheduler(clip_sample_range=3.0); sch.set_timesteps(250)
            lat = sampling.sample(m32, x1.cuda(), crf1.cuda(), crl1.cuda(), sch).cpu()
            ps, rng = psnr_pp(lat, gf["final"])
            say(f"latent 32, full 250-step DDIM (B=1) vs reference golden: PSNR {ps:.1f} dB on peak-to-peak {rng:.2f}, rel-L2 {rel_l2(lat, gf['final']):.3e}")
        del m32
    if "image" in what and "full" in what and os.path.exists(os.path.join(ROOT, "tests", "golden", "ddpm1000_L16.npz")):
        from hifidiff_amd.vae import AutoencoderKL
        PV = synth.vae_state_dict()
        vae = AutoencoderKL(); vae.load_state_dict(PV); vae.to("cuda:0")
        img_hip = vae.decode_scaled(final_hip.cuda()).cpu()
        img_ref = O.vae_decode_scaled(PV, T(golden("ddpm1000_L16.npz")["final"]))        # fp32 oracle decode of the REFERENCE network's latent
        ps, rng = psnr_pp(img_hip, img_ref)
        say(f"image space (synthetic-weight VAE, parity unpinned): HIP decode(HIP 1000-step latent) vs fp32 oracle decode(reference latent): PSNR {ps:.1f} dB on "
            f"peak-to-peak {rng:.2f}, rel-L2 {rel_l2(img_hip, img_ref):.3e}")
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
        f.write("\n".join(OUT) + "\n")


if __name__ == "__main__":
    main()
