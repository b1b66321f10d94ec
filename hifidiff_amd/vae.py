"""`AutoencoderKL` mirror for the VAE boundary either side of the sampling loop:

    cr_latent = vae.encode(F.interpolate(cr_face, image_res, mode="bicubic")).latent_dist.sample() * 0.18215   test_refiner.py:78-83
    images    = vae.decode(latent / 0.18215).sample                                                              test_refiner.py:93

The reference uses diffusers' `AutoencoderKL.from_pretrained("Manojb/stable-diffusion-2-1-base", subfolder="vae")`
(test_refiner.py:176-178): third-party code and weights, neither present offline.  This module keeps the part of that
surface the reference touches (`encode(x).latent_dist.sample()/.mode()/.mean/.logvar`, `decode(z).sample`,
`load_state_dict` with diffusers' key names, `.to(device)`, `.config.scaling_factor`), and runs the network in
libhifidiff_hip.so (hd_vae_encode / hd_vae_decode).  No CPU path.  `encode_scaled` / `decode_scaled` are the fused forms
(bicubic resize, posterior sample and the 0.18215 factor inside the library)."""
import ctypes

import torch
from torch import nn

from . import _lib, arch


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution over moments [B,8,L,L]."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        z = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * z

    def mode(self):
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class _Config:
    scaling_factor = arch.VAE_SCALING
    latent_channels = 4
    in_channels = 3
    out_channels = 3
    block_out_channels = arch.VAE_CHANNELS


class AutoencoderKL(nn.Module):
    def __init__(self):
        super().__init__()
        self.config = _Config()
        self._ctx, self._device, self._state, self._loaded, self._keep = None, None, None, False, None

    @classmethod
    def from_pretrained(cls, name_or_path, subfolder=None, **kw):
        """A local diffusers directory (config + diffusion_pytorch_model.safetensors) only: there is no hub access here."""
        import os
        path = os.path.join(name_or_path, subfolder) if subfolder else name_or_path
        f = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if not os.path.exists(f):
            raise OSError("AutoencoderKL.from_pretrained needs a local directory holding diffusion_pytorch_model.safetensors "
                          "(no hub access); got %s" % path)
        from safetensors.torch import load_file
        m = cls()
        m.load_state_dict(load_file(f))
        return m

    # ---- state dict plumbing ----
    # diffusers < 0.17 checkpoints (the SD-2.x era VAE files among them) name the mid-block attention projections
    # query / key / value / proj_attn; diffusers 0.32.2 renames them at load time (_convert_deprecated_attention_blocks)
    _DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}

    @classmethod
    def _canonical_keys(cls, sd, man):
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if len(parts) >= 3 and parts[-2] in cls._DEPRECATED_ATTN and "attentions" in parts:
                k = ".".join(parts[:-2] + [cls._DEPRECATED_ATTN[parts[-2]], parts[-1]])
            if k not in man and not torch.is_tensor(v):
                continue                                   # non-weight entries (metadata some exporters add)
            out[k] = v
        return out

    def load_state_dict(self, sd, strict=True):
        man = arch.vae_manifest()
        sd = self._canonical_keys(sd, man)
        missing = [k for k in man if k not in sd]
        unexpected = [k for k in sd if k not in man]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for AutoencoderKL: Missing key(s): %s; Unexpected key(s): %s"
                               % (missing[:4], unexpected[:4]))
        for k, (shape, _, _) in man.items():
            if k in sd:
                got = tuple(sd[k].shape)
                if got != tuple(shape) and not (len(shape) == 2 and got == tuple(shape) + (1, 1)):   # older checkpoints store the attention Linear as 1x1 conv
                    raise RuntimeError("size mismatch for %s: got %s, expected %s" % (k, got, tuple(shape)))
        self._state = {k: sd[k].detach().reshape(man[k][0]) for k in man if k in sd}
        if self._ctx is not None:
            self._upload()                                 # a context whose weights are finalized is replaced (self._loaded)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def state_dict(self, *a, **k):
        return dict(self._state or {})

    def to(self, *args, **kwargs):
        device = kwargs.get("device", args[0] if args else None)
        if isinstance(device, (str, torch.device, int)):
            self._ensure(torch.device("cuda", device) if isinstance(device, int) else torch.device(device))
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def _ensure(self, device):
        if device.type != "cuda":
            raise RuntimeError("hifidiff_amd runs on an MI355X (gfx950) GPU only; got device %s (no CPU fallback)" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._ctx is not None:
            if idx != self._device.index:
                raise RuntimeError("this model already lives on cuda:%d" % self._device.index)
            return
        ctx = ctypes.c_void_p()
        _lib.check(_lib.lib().hd_vae_create(ctypes.byref(ctx), idx))
        self._ctx, self._device = ctx, torch.device("cuda", idx)
        if self._state is not None:
            self._upload()

    def _upload(self):
        man = arch.vae_manifest()
        if any(k not in self._state for k in man):
            raise RuntimeError("state dict incomplete: %d of %d tensors loaded" % (len(self._state), len(man)))
        L = _lib.lib()
        keep, descs = [], (_lib.TensorDesc * len(man))()
        for i, k in enumerate(man):
            t = self._state[k].to(torch.float32).contiguous()
            keep.append(t)
            d = descs[i]
            d.name, d.data, d.ndim, d.is_device = k.encode(), t.data_ptr(), t.dim(), 1 if t.is_cuda else 0
            for j, s in enumerate(t.shape):
                d.shape[j] = s
        if self._loaded:
            L.hd_destroy(self._ctx)
            self._ctx = None
            self._loaded = False                             # nothing usable until finalize has succeeded
            ctx = ctypes.c_void_p()
            _lib.check(L.hd_vae_create(ctypes.byref(ctx), self._device.index))
            self._ctx = ctx
        with torch.cuda.device(self._device):
            _lib.check(L.hd_load_weights(self._ctx, descs, len(man)), self._ctx)
            _lib.check(L.hd_finalize_weights(self._ctx), self._ctx)
        self._loaded = True

    def _ready(self, x):
        self._ensure(x.device)
        if not self._loaded:
            raise RuntimeError("weights are not loaded: call load_state_dict(...) and move the model to a cuda device")

    # ---- the reference's calls ----
    def _encode(self, x, image_res, vae_range, want_moments, noise, seed):
        self._ready(x)
        B = x.shape[0]
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
            raise RuntimeError("AutoencoderKL.encode input must be (B,3,R,R), got %s" % (tuple(x.shape),))
        R = int(image_res) if image_res is not None else int(x.shape[2])
        if R % 64 or not 64 <= R <= 512:
            raise RuntimeError("image size must be a multiple of 64 in [64, 512], got %d" % R)
        Lr = R // 8
        if B == 0:
            return torch.empty((0, 8 if want_moments else 4, Lr, Lr), dtype=torch.float32, device=self._device)
        xin = x.to(device=self._device, dtype=torch.float32).contiguous()
        out = torch.empty((B, 8 if want_moments else 4, Lr, Lr), dtype=torch.float32, device=self._device)
        nz = None
        if noise is not None:
            nz = noise.to(device=self._device, dtype=torch.float32).contiguous()
            if tuple(nz.shape) != (B, 4, Lr, Lr):
                raise RuntimeError("noise must be (B,4,%d,%d)" % (Lr, Lr))
        with torch.cuda.device(self._device):
            _lib.check(_lib.lib().hd_vae_encode(self._ctx, B, int(x.shape[2]), R, xin.data_ptr(), 1 if vae_range else 0,
                                                nz.data_ptr() if nz is not None else None, int(seed),
                                                out.data_ptr() if want_moments else None, None if want_moments else out.data_ptr(),
                                                torch.cuda.current_stream(self._device).cuda_stream), self._ctx)
        self._keep = (xin, nz, out)                  # the launch program holds these pointers until the next call
        return out

    def encode(self, x, return_dict=True):
        """`vae.encode(x).latent_dist` (diffusers AutoencoderKL.encode): the posterior over moments [B,8,L,L]."""
        post = DiagonalGaussianDistribution(self._encode(x, None, False, True, None, 0).clone())
        return AutoencoderKLOutput(post) if return_dict else (post,)

    def encode_scaled(self, images, image_res, vae_range=False, noise=None, seed=0):
        """cr_latent of test_refiner.py:78-83 in one call: bicubic to image_res, encode, posterior sample (noise tensor or device
        Philox(seed)), times 0.18215.  vae_range=True applies train_refiner.py's to_vae_range first."""
        return self._encode(images, image_res, vae_range, False, noise, seed).clone()

    def decode(self, z, return_dict=True):
        """`vae.decode(z).sample` (z is the unscaled latent, as the reference passes latent / 0.18215)."""
        out = self.decode_scaled(z * self.config.scaling_factor)
        return DecoderOutput(out) if return_dict else (out,)

    def decode_scaled(self, latents):
        """images = vae.decode(latents / 0.18215).sample with the division inside the library."""
        self._ready(latents)
        B = latents.shape[0]
        if latents.dim() != 4 or latents.shape[1] != 4 or latents.shape[2] != latents.shape[3] or latents.shape[2] % 8:
            raise RuntimeError("AutoencoderKL.decode input must be (B,4,L,L) with L a multiple of 8, got %s" % (tuple(latents.shape),))
        Lr = int(latents.shape[2])
        if B == 0:
            return torch.empty((0, 3, 8 * Lr, 8 * Lr), dtype=torch.float32, device=self._device)
        zin = latents.to(device=self._device, dtype=torch.float32).contiguous()
        out = torch.empty((B, 3, 8 * Lr, 8 * Lr), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(_lib.lib().hd_vae_decode(self._ctx, B, Lr, zin.data_ptr(), out.data_ptr(),
                                                torch.cuda.current_stream(self._device).cuda_stream), self._ctx)
        self._keep = (zin, out)
        return out.clone()

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.lib().hd_destroy(self._ctx)
        except Exception:
            pass
