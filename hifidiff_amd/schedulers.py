"""DDIMScheduler / DDPMScheduler with the surface the reference's sampling loop uses
(test_refiner.py:85-91,166-171; train_refiner.py:109-120,337-348):
    sched = DDIMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear",
                          prediction_type="epsilon", clip_sample_range=3.0)
    sched.set_timesteps(50); for t in sched.timesteps: x = sched.step(eps, t, x, eta=0.0).prev_sample
The arithmetic restates diffusers 0.32.2 (not installed here; "parity unpinned"): the
per-step scalars are computed on the host in fp32 exactly in diffusers' order, the elementwise update
runs in the HIP kernel `sched_step_direct_kernel` (or inside the captured graph for `sample()`).
"""
import ctypes

import torch

from . import _lib


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _Base:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="scaled_linear",
                 prediction_type="epsilon", clip_sample=True, clip_sample_range=1.0, **unused):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon":
            raise NotImplementedError("the reference only uses beta_schedule='scaled_linear', prediction_type='epsilon'")
        self.num_train_timesteps = int(num_train_timesteps)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, self.num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.clip_sample = bool(clip_sample)
        self.clip_sample_range = float(clip_sample_range)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = self.num_train_timesteps
        self.timesteps = torch.arange(self.num_train_timesteps - 1, -1, -1, dtype=torch.long)
        self._timesteps_set = False                        # diffusers' num_inference_steps is None until set_timesteps

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        r = self.num_train_timesteps // num_inference_steps
        self.num_inference_steps = int(num_inference_steps)
        self._timesteps_set = True
        self.timesteps = (torch.arange(0, num_inference_steps, dtype=torch.long) * r).flip(0)   # 'leading', offset 0
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(abar_t) x0 + sqrt(1-abar_t) n (train_refiner.py:168); plain torch plumbing, not on the sampling path."""
        a = self.alphas_cumprod.to(original_samples.device)[timesteps.long()].to(original_samples.dtype)
        sa = (a ** 0.5).view(-1, *([1] * (original_samples.dim() - 1)))
        sb = ((1 - a) ** 0.5).view(-1, *([1] * (original_samples.dim() - 1)))
        return sa * original_samples + sb * noise

    def _alpha(self, t):
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod

    def _coef(self, t):
        raise NotImplementedError

    def coefficient_table(self):
        """(timesteps [n] fp32, coef [n,7] fp32) for hd_sample: see hd_schedule in include/hifidiff_hip.h.
        The table depends on the schedule only (1000 x a dozen fp32 scalar operations in diffusers' order: 12 ms of host time),
        so it is kept until anything `_coef` reads changes (timesteps, step counts, clipping, the alpha tables) -- like the FiLM
        table of the schedule inside hd_sample.  The two tensors are shared with the cache: read-only for the caller."""
        ts = [int(t) for t in (self.timesteps.tolist() if hasattr(self.timesteps, "tolist") else self.timesteps)]
        ac, fa = self.alphas_cumprod, self.final_alpha_cumprod
        # everything _coef reads: the schedule, the clipping, and the alpha tables (identity + in-place version of the tensors)
        key = (tuple(ts), self.num_inference_steps, self.num_train_timesteps, self.clip_sample, self.clip_sample_range,
               id(ac), getattr(ac, "_version", 0), float(fa))
        hit = getattr(self, "_coef_cache", None)
        if hit is None or hit[0] != key:
            hit = (key, torch.tensor(ts, dtype=torch.float32),
                   torch.tensor([self._coef(t) for t in ts], dtype=torch.float32).reshape(len(ts), 7))
            self._coef_cache = hit
        return hit[1], hit[2]

    def _launch(self, eps, t, x, noise, seed, step):
        if not (x.is_cuda and eps.is_cuda):
            raise RuntimeError("hifidiff_amd schedulers run on the GPU only (no CPU fallback)")
        if eps.device != x.device or (noise is not None and noise.is_cuda and noise.device != x.device):
            raise RuntimeError("scheduler.step: sample, model_output and noise must live on one device (%s vs %s)" % (x.device, eps.device))
        x = x.contiguous().clone()
        eps = eps.contiguous().to(torch.float32)
        c = (ctypes.c_float * 7)(*self._coef(int(t)))
        nptr = None
        if noise is not None:
            noise = noise.contiguous().to(device=x.device, dtype=torch.float32)
            nptr = noise.data_ptr()
        with torch.cuda.device(x.device):                  # the C function takes no context: launch on x's device and stream
            rc = _lib.lib().hd_scheduler_step(x.data_ptr(), eps.data_ptr(), c, nptr, int(seed), int(step), x.numel(),
                                              torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc)
        return SchedulerOutput(x)


class DDIMScheduler(_Base):
    def _coef(self, t):
        r = self.num_train_timesteps // self.num_inference_steps
        a, a_p = self._alpha(t), self._alpha(t - r)
        clip = self.clip_sample_range if self.clip_sample else float("inf")
        return [float((1 - a) ** 0.5), float(a ** 0.5), clip, float(a_p ** 0.5), 0.0, float((1 - a_p) ** 0.5), 0.0]

    def step(self, model_output, timestep, sample, eta=0.0, **kw):
        if eta != 0.0:
            raise NotImplementedError("only eta=0 (the reference's setting, test_refiner.py:91) is implemented")
        if not self._timesteps_set:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        eps = getattr(model_output, "sample", model_output)
        return self._launch(eps, timestep, sample, None, 0, 0)


class DDPMScheduler(_Base):
    """variance_type 'fixed_small'.  `step(..., noise=z)` takes the noise explicitly; without it z is
    drawn from Philox(seed, step_index) on the device."""

    def _coef(self, t):
        r = self.num_train_timesteps // self.num_inference_steps
        a, a_p = self._alpha(t), self._alpha(t - r)
        alpha_t = a / a_p
        beta_t = 1 - alpha_t
        clip = self.clip_sample_range if self.clip_sample else float("inf")
        sig = float(torch.clamp((1 - a_p) / (1 - a) * beta_t, min=1e-20) ** 0.5) if t > 0 else 0.0
        return [float((1 - a) ** 0.5), float(a ** 0.5), clip, float(a_p ** 0.5 * beta_t / (1 - a)),
                float(alpha_t ** 0.5 * (1 - a_p) / (1 - a)), 0.0, sig]

    def step(self, model_output, timestep, sample, noise=None, seed=0, **kw):
        eps = getattr(model_output, "sample", model_output)
        ts = [int(v) for v in self.timesteps]
        if int(timestep) not in ts:                        # the Philox stream is keyed by the step's index in the schedule
            raise ValueError("timestep %d is not in the current schedule (set_timesteps changed, or a stale t)" % int(timestep))
        return self._launch(eps, timestep, sample, noise, seed, ts.index(int(timestep)))
