"""The reverse-diffusion loop of the refiner in latent space.

`ddim_sample_eager` is the body of the reference's `ddim_sample` (test_refiner.py:85-91,
train_refiner.py:109-120) verbatim against the mirrored modules: one `model(...)` + one
`scheduler.step(...)` per Python iteration.

`sample` is the MI355X-first form of the same loop: conditioning once, FiLM table for all steps once,
then one captured hipGraph replayed n_steps times inside libhifidiff_hip.so (`hd_sample`).
"""
import ctypes

import torch

from . import _lib


@torch.no_grad()
def ddim_sample_eager(model, latents, cr_face, cr_latent, scheduler, num_inference_steps=50):
    bs = latents.shape[0]
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    for t in scheduler.timesteps:
        t_batch = torch.full((bs,), int(t), device=latents.device, dtype=torch.long)
        noise_pred = model(latents, t_batch, cr_face, cr_latent).sample
        latents = scheduler.step(noise_pred, t, latents, eta=0.0).prev_sample
    return latents


@torch.no_grad()
def ddim_sample_eager_unconditional(model, latents, scheduler, num_inference_steps=50):
    """The unconditional loop of pretrain_denoiser.py:99-110 against the mirrored `Denoiser`."""
    bs = latents.shape[0]
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    for t in scheduler.timesteps:
        t_batch = torch.full((bs,), int(t), device=latents.device, dtype=torch.long)
        noise_pred = model(latents, t_batch).sample
        latents = scheduler.step(noise_pred, t, latents, eta=0.0).prev_sample
    return latents


@torch.no_grad()
def sample(model, latents, cr_face, cr_latent, scheduler, noise=None, seed=0, prepare=True, check=True):
    """Whole loop on the GPU: returns the final latents (a new tensor).

    noise: optional [n_steps, B, 4, L, L] tensor of z (DDPM); None -> device Philox(seed).
    For the unconditional `Denoiser` pass cr_face = cr_latent = None.
    check=True (the default): ONE stream synchronisation after the whole loop (not per step), then RuntimeError if a persistent
    stage launch gave up during it -- where the reference's loop would have raised (test_refiner.py:89-91), so that the last batch
    of a val_loop cannot end with rc 0 and NaN images.  check=False only enqueues the work (the returned latents are NaN in the
    failing case either way; `model.check()` reports it later).  bench.py times the loop with its own synchronisation."""
    e = model.engine
    e.ensure(latents.device)
    if latents.shape[0] == 0:                              # empty batch: nothing to sample
        return latents.to(device=e.device, dtype=torch.float32).clone()
    if not e.conditional:
        if cr_face is not None or cr_latent is not None:
            raise RuntimeError("the unconditional Denoiser takes no cr_face / cr_latent")
        e.require_loaded()
        e.prepare_unconditional(latents.shape[0])
    elif prepare:
        model.prepare(cr_face, cr_latent)
    e.require_loaded()
    x = latents.to(device=e.device, dtype=torch.float32).contiguous().clone()
    ts, coef = scheduler.coefficient_table()
    ts, coef = ts.contiguous(), coef.contiguous()
    sch = _lib.Schedule()
    sch.n_steps = ts.numel()
    sch.timesteps = ctypes.cast(ts.data_ptr(), ctypes.POINTER(ctypes.c_float))
    sch.coef = ctypes.cast(coef.data_ptr(), ctypes.POINTER(ctypes.c_float))
    nptr = None
    if noise is not None:
        noise = noise.to(device=e.device, dtype=torch.float32).contiguous()
        if noise.numel() != ts.numel() * x.numel():
            raise RuntimeError("noise must be [n_steps, B, 4, L, L]")
        nptr = noise.data_ptr()
    with torch.cuda.device(e.device):
        _lib.check(_lib.lib().hd_sample(e.ctx, x.data_ptr(), ctypes.byref(sch), nptr, int(seed),
                                        torch.cuda.current_stream(e.device).cuda_stream), e.ctx)
    if check:
        e.check()
    return x
