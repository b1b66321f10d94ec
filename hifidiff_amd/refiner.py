"""Drop-in mirror of the reference's module surface for the refiner sampling path.

    FacialRefiner(latent_res=16, idc_ckpt=None, denoiser_ckpt=None)   models/refiner.py:10-16
        .forward(latents, timesteps, cr_face, cr_latent) -> UNet2DOutput    models/refiner.py:32-38
        .idc / .denoiser / .fpg sub-modules                                   models/refiner.py:14-16
    FusedDenoiser(latent_size)                                               models/denoiser/model.py:137
        .forward(latents, timesteps, facial_priors, identity_embedding)      models/denoiser/model.py:217
        .config.in_channels / .config.sample_size / .dtype / .width          models/denoiser/model.py:141-146
    UNet2DOutput(.sample)                                                    models/denoiser/model.py:11-13

Same constructor arguments, forward signatures, state-dict keys and error behaviour (RuntimeError on
bad shapes / missing keys), so the bodies of `ddim_sample` (test_refiner.py:58-95,
train_refiner.py:86-125) run unchanged.  All compute happens in libhifidiff_hip.so (hand-written HIP
for gfx950); PyTorch only owns the tensors and the stream.  There is no CPU path.

Deviation that is an optimisation, not a semantic change: the reference recomputes `fpg(cr_latent)`
and `idc(cr_face)` on every forward although they are step-invariant (refiner.py:33-34); here the
conditioning of a (cr_face, cr_latent) pair is computed once and reused while the SAME tensor objects
(`is`, with unchanged version counters) are passed again -- the loop of `ddim_sample` passes the same
two objects on every step.  The cache holds strong references to both tensors, so their storage cannot
be recycled for another batch while the key is live; nothing is ever inferred from addresses.
`cache_conditioning=False` restores as-written behaviour.

One model instance serves any batch size (the ragged last batch of the reference's `val_loop`,
test_refiner.py:98-112,160): the library keeps a workspace per recent batch size next to the shared
packed weights.
"""
import ctypes

import torch
from torch import nn

from . import _lib, arch


class UNet2DOutput:
    def __init__(self, data):
        self.sample = data


class _Config:
    pass


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _version(t):
    """Version counter of a tensor, or None for tensors that do not track one (created under torch.inference_mode():
    reading `_version` raises there).  None never matches a cached key, so such tensors are re-prepared on every call."""
    try:
        return t._version
    except RuntimeError:
        return None


class _Engine:
    """Owns the hd_ctx of one (latent_res, device)."""

    def __init__(self, latent_res, conditional=True):
        self.latent_res = int(latent_res)
        self.conditional = conditional     # False: the unconditional Denoiser (models/denoiser/model.py:32)
        self.ctx = None
        self.device = None
        self.state = None          # CPU copy of the loaded state dict (for state_dict())
        self.loaded = False
        self.cond_key = None
        self.batch = None

    def ensure(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("hifidiff_amd runs on an MI355X (gfx950) GPU only; got device %s (no CPU fallback)" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self.ctx is not None:
            if idx != self.device.index:
                raise RuntimeError("this model already lives on cuda:%d" % self.device.index)
            return
        ctx = self._create(idx)
        self.ctx, self.device = ctx, torch.device("cuda", idx)
        if self.state is not None:
            self._upload()

    def _create(self, idx):
        L = _lib.lib()
        ctx = ctypes.c_void_p()
        _lib.check((L.hd_create if self.conditional else L.hd_create_unconditional)(ctypes.byref(ctx), self.latent_res, idx))
        return ctx

    def manifest(self):
        if self.conditional:
            return arch.refiner_manifest(self.latent_res)
        return arch.denoiser_manifest(self.latent_res, prefix="denoiser", fused=False)

    def load(self, sd, strict=True):
        man = self.manifest()
        missing = [k for k in man if k not in sd]
        unexpected = [k for k in sd if k not in man]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for %s: Missing key(s): %s; Unexpected key(s): %s"
                               % ("FacialRefiner" if self.conditional else "Denoiser", missing[:4], unexpected[:4]))
        for k, (shape, _, _) in man.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: got %s, expected %s" % (k, tuple(sd[k].shape), tuple(shape)))
        base = self.state or {}
        self.state = {k: (sd[k] if k in sd else base[k]).detach() for k in man if (k in sd or k in base)}
        if self.ctx is not None:
            self._upload()                                 # a context whose weights are finalized is replaced (self.loaded)
        return missing, unexpected

    def _upload(self):
        man = self.manifest()
        if any(k not in self.state for k in man):
            raise RuntimeError("state dict incomplete: %d of %d tensors loaded" % (len(self.state), len(man)))
        L = _lib.lib()
        keep, descs = [], (_lib.TensorDesc * len(man))()
        for i, k in enumerate(man):
            t = self.state[k]
            if t.dtype != torch.int64:
                t = t.to(torch.float32)
            t = t.contiguous()
            keep.append(t)
            d = descs[i]
            d.name = k.encode()
            d.data = t.data_ptr()
            d.ndim = t.dim()
            for j, s in enumerate(t.shape):
                d.shape[j] = s
            d.is_device = 1 if t.is_cuda else 0
        if self.loaded:      # re-load: a fresh context is simplest (weights are packed once)
            L.hd_destroy(self.ctx)
            self.ctx = None
            self.loaded, self.cond_key, self.batch, self.prior_key = False, None, None, None      # nothing usable until finalize succeeds
            self.ctx = self._create(self.device.index)
        with torch.cuda.device(self.device):
            _lib.check(L.hd_load_weights(self.ctx, descs, len(man)), self.ctx)
            _lib.check(L.hd_finalize_weights(self.ctx), self.ctx)
        self.loaded, self.cond_key, self.batch, self.prior_key = True, None, None, None

    def after_submodule_call(self, batch):
        """`model.fpg(x)` / `model.idc(x)` run on the workspace of THEIR batch size: with another batch than the prepared one
        the library has parked the prepared workspace (its conditioning is gone), and hd_fpg overwrites the priors of the
        active one in any case -- the cached conditioning of a following forward() must not be trusted."""
        if batch != self.batch:
            self.batch = None
        self.cond_key = None
        self.prior_key = None

    def require_loaded(self):
        if self.ctx is None or not self.loaded:
            raise RuntimeError("weights are not loaded: call load_state_dict(...) and move the model to a cuda device")

    # ---- C-ABI calls ----
    def prepare(self, cr_latent, cr_face=None, id_emb=None):
        self.require_loaded()
        B = cr_latent.shape[0]
        L = self.latent_res
        if tuple(cr_latent.shape) != (B, 4, L, L):
            raise RuntimeError("cr_latent must be (B,4,%d,%d), got %s" % (L, L, tuple(cr_latent.shape)))
        if cr_face is not None and tuple(cr_face.shape) != (B, 3, 128, 128):
            raise RuntimeError("cr_face must be (B,3,128,128), got %s" % (tuple(cr_face.shape),))
        crl = _f32c(cr_latent, self.device)
        crf = _f32c(cr_face, self.device) if cr_face is not None else None
        emb = _f32c(id_emb.reshape(B, -1), self.device) if id_emb is not None else None
        if emb is not None and emb.shape[1] != 2048:
            raise RuntimeError("identity embedding must have 2048 features")
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hd_prepare(self.ctx, B, crl.data_ptr(), crf.data_ptr() if crf is not None else None,
                                             emb.data_ptr() if emb is not None else None, _stream(self.device)), self.ctx)
        self.batch, self.cond_key, self.prior_key = B, None, None

    def prepare_unconditional(self, batch):
        self.require_loaded()
        if self.batch != batch:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().hd_prepare_unconditional(self.ctx, batch, _stream(self.device)), self.ctx)
            self.batch, self.cond_key = batch, None

    def prepare_from_priors(self, priors, id_emb):
        self.require_loaded()
        B = priors[0].shape[0]
        s = self.latent_res // 16
        if len(priors) != 5:
            raise RuntimeError("facial_priors must hold 5 maps")
        keep = []
        for i, p in enumerate(priors):
            want = (B, 2048 >> i, s << i, s << i)
            if tuple(p.shape) != want:
                raise RuntimeError("facial_priors[%d] must be %s, got %s" % (i, want, tuple(p.shape)))
            keep.append(_f32c(p, self.device))
        emb = _f32c(id_emb.reshape(B, -1), self.device)
        ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in keep])
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hd_prepare_from_priors(self.ctx, B, ptrs, emb.data_ptr(), _stream(self.device)), self.ctx)
        self.batch, self.cond_key = B, None

    def timesteps_tensor(self, timesteps, batch):
        """Scalar / 0-d / (1,) / (B,) int or float -> fp32 device tensor of 1 or B values (model.py:218-229)."""
        if isinstance(timesteps, (int, float)):
            return torch.full((1,), float(timesteps), dtype=torch.float32, device=self.device)
        t = torch.as_tensor(timesteps)
        if t.dim() == 0:
            t = t.reshape(1)
        if t.dim() != 1 or t.shape[0] not in (1, batch):
            raise RuntimeError("timesteps must be a scalar or have shape (1,) or (%d,), got %s" % (batch, tuple(t.shape)))
        return _f32c(t, self.device)

    def eps(self, latents, timesteps):
        self.require_loaded()
        B, L = latents.shape[0], self.latent_res
        if tuple(latents.shape) != (B, 4, L, L):
            raise RuntimeError("latents must be (B,4,%d,%d), got %s" % (L, L, tuple(latents.shape)))
        if self.batch != B:
            raise RuntimeError("conditioning was prepared for batch %s, latents have batch %d" % (self.batch, B))
        x = _f32c(latents, self.device)
        t = self.timesteps_tensor(timesteps, B)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hd_eps(self.ctx, x.data_ptr(), t.data_ptr(), t.numel(), out.data_ptr(), _stream(self.device)), self.ctx)
        return out

    def check(self, synchronize=True):
        """Where the reference's loop would have raised synchronously (test_refiner.py:89-91): forward() / sample() only enqueue
        work, and a persistent stage launch that has to give up (another tenant kept one of its workgroups off the GPU) fills the
        result of THAT call with NaN on the device.  This synchronises the current stream and raises RuntimeError once for such
        a call; the context has then switched itself to one launch per GEMM and the next call is valid again.  EVERY call issued
        between the stage giving up and this check may be poisoned (a call enqueued behind the failing one runs its stages before
        the host has seen the word): the error is raised once and names the first failure."""
        if self.ctx is None:
            return
        if synchronize:
            torch.cuda.current_stream(self.device).synchronize()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().hd_check(self.ctx), self.ctx)

    def __del__(self):
        try:
            if self.ctx is not None:
                _lib.lib().hd_destroy(self.ctx)
        except Exception:
            pass


class _SubModule(nn.Module):
    def __init__(self, engine):
        super().__init__()
        object.__setattr__(self, "_engine", engine)


class ResNet50(_SubModule):
    """`model.idc`: (B,3,128,128) -> (B,2048,1,1) (models/idc/model.py:122-135)."""

    def forward(self, x):
        e = self._engine
        e.ensure(x.device); e.require_loaded()
        B = x.shape[0]
        if tuple(x.shape) != (B, 3, 128, 128):
            raise RuntimeError("ResNet50 input must be (B,3,128,128), got %s" % (tuple(x.shape),))
        xin = _f32c(x, e.device)
        out = torch.empty((B, 2048), dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().hd_idc(e.ctx, B, xin.data_ptr(), out.data_ptr(), _stream(e.device)), e.ctx)
        e.after_submodule_call(B)
        return out.reshape(B, 2048, 1, 1)


class FacialPriorGuidance(_SubModule):
    """`model.fpg`: (B,4,L,L) -> 5 prior maps, coarsest first (models/fpg/model.py:46-64)."""

    def forward(self, x):
        e = self._engine
        e.ensure(x.device); e.require_loaded()
        B, L = x.shape[0], e.latent_res
        if tuple(x.shape) != (B, 4, L, L):
            raise RuntimeError("FacialPriorGuidance input must be (B,4,%d,%d), got %s" % (L, L, tuple(x.shape)))
        xin = _f32c(x, e.device)
        s = L // 16
        outs = [torch.empty((B, 2048 >> i, s << i, s << i), dtype=torch.float32, device=e.device) for i in range(5)]
        ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in outs])
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().hd_fpg(e.ctx, B, xin.data_ptr(), ptrs, _stream(e.device)), e.ctx)
        e.after_submodule_call(B)
        return outs


class FusedDenoiser(_SubModule):
    def __init__(self, latent_size, _engine=None):
        super().__init__(_engine if _engine is not None else _Engine(latent_size))
        self.width = 32 * 4
        self.dtype = torch.float32
        self.config = _Config()
        self.config.in_channels = 4
        self.config.sample_size = latent_size

    def forward(self, latents, timesteps, facial_priors, identity_embedding):
        e = self._engine
        e.ensure(latents.device)
        # the gates and idc_conv depend on the priors / embedding only (models/fpg/hca.py:26-27, models/denoiser/model.py:245):
        # the loop passes the same objects every step -> computed once (identity + version, strong references: never addresses)
        k = getattr(e, "prior_key", None)
        objs = list(facial_priors) + [identity_embedding]
        # (tensors without a version counter -- torch.inference_mode() -- are never a hit; writes through `.data`, numpy or
        # DLPack aliases do not bump the counter: pass a new tensor object or call invalidate_conditioning() after such a write)
        hit = (k is not None and len(k) == len(objs) and e.batch == latents.shape[0] and e.cond_key is None and
               all(o is ko and kv is not None and _version(o) == kv for o, (ko, kv) in zip(objs, k)))
        if not hit:
            e.prior_key = None
            e.prepare_from_priors(facial_priors, identity_embedding)
            e.prior_key = [(o, _version(o)) for o in objs]
        return UNet2DOutput(e.eps(latents, timesteps))

    def invalidate_conditioning(self):
        """Forget the cached gates / idc term (after writing into the prior tensors through an alias that does not bump
        their version counter)."""
        self._engine.prior_key = None


class Denoiser(nn.Module):
    """The unconditional pre-training network (models/denoiser/model.py:32-134): `model(latents, t).sample`, as the
    sampling loop of pretrain_denoiser.py:101-110 calls it.  State-dict keys are the reference's (no prefix)."""

    def __init__(self, latent_size):
        super().__init__()
        if latent_size % 16 != 0 or latent_size < 16:
            raise ValueError("latent_size must be a multiple of 16")
        object.__setattr__(self, "_engine", _Engine(latent_size, conditional=False))
        self.width = 32 * 4
        self.dtype = torch.float32
        self.config = _Config()
        self.config.in_channels = 4
        self.config.sample_size = latent_size

    def load_state_dict(self, state_dict, strict=True):
        missing, unexpected = self._engine.load({"denoiser." + k: v for k, v in state_dict.items()}, strict)
        n = len("denoiser.")
        return torch.nn.modules.module._IncompatibleKeys([k[n:] for k in missing], [k[n:] for k in unexpected])

    def state_dict(self, *a, **k):
        n = len("denoiser.")
        return {key[n:]: v for key, v in (self._engine.state or {}).items()}

    def to(self, *args, **kwargs):
        device = kwargs.get("device", args[0] if args else None)
        if isinstance(device, (str, torch.device, int)):
            self._engine.ensure(torch.device("cuda", device) if isinstance(device, int) else device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    @property
    def engine(self):
        return self._engine

    def check(self, synchronize=True):
        """Synchronise and raise RuntimeError if a call since the last check handed back NaN-poisoned results (_Engine.check)."""
        self._engine.check(synchronize)

    def forward(self, latents, timesteps):
        e = self._engine
        e.ensure(latents.device)
        if latents.shape[0] == 0:
            return UNet2DOutput(torch.empty_like(latents, dtype=torch.float32))
        e.prepare_unconditional(latents.shape[0])
        return UNet2DOutput(e.eps(latents, timesteps))


class FacialRefiner(nn.Module):
    def __init__(self, latent_res=16, idc_ckpt=None, denoiser_ckpt=None, cache_conditioning=True):
        super().__init__()
        if latent_res % 16 != 0 or latent_res < 16:
            raise ValueError("latent_res must be a multiple of 16")
        object.__setattr__(self, "_engine", _Engine(latent_res))
        self.latent_res = latent_res
        self.cache_conditioning = cache_conditioning
        self.idc = ResNet50(self._engine)
        self.denoiser = FusedDenoiser(latent_res, self._engine)
        self.fpg = FacialPriorGuidance(self._engine)
        if idc_ckpt is not None or denoiser_ckpt is not None:
            # models/refiner.py:18-25 initialises from an IDC .pt and a denoiser .safetensors
            sd = {}
            if idc_ckpt is not None:
                sd.update({"idc." + k: v for k, v in torch.load(idc_ckpt, map_location="cpu")["model_state_dict"].items()})
            if denoiser_ckpt is not None:
                from safetensors.torch import load_file
                w = load_file(denoiser_ckpt)
                man = arch.refiner_manifest(latent_res)
                sd.update({"denoiser." + k: v for k, v in w.items() if "denoiser." + k in man})
                sd.update({"fpg." + k: v for k, v in w.items() if "fpg." + k in man})
            self._engine.load(sd, strict=False)

    # ---- nn.Module plumbing ----
    def load_state_dict(self, state_dict, strict=True):
        missing, unexpected = self._engine.load(state_dict, strict)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def state_dict(self, *a, **k):
        return dict(self._engine.state or {})

    def to(self, *args, **kwargs):
        device = kwargs.get("device", args[0] if args else None)
        if isinstance(device, (str, torch.device, int)):
            self._engine.ensure(torch.device("cuda", device) if isinstance(device, int) else device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    @property
    def engine(self):
        return self._engine

    def check(self, synchronize=True):
        """Synchronise and raise RuntimeError if a call since the last check handed back NaN-poisoned results (_Engine.check)."""
        self._engine.check(synchronize)

    def prepare(self, cr_face, cr_latent):
        """Once-per-batch conditioning (FPG, IDC, HCA gates, idc_conv)."""
        e = self._engine
        e.ensure(cr_latent.device)
        B, L = cr_latent.shape[0], e.latent_res
        if tuple(cr_latent.shape) != (B, 4, L, L) or tuple(cr_face.shape) != (B, 3, 128, 128):
            raise RuntimeError("expected cr_latent (B,4,%d,%d) and cr_face (B,3,128,128), got %s and %s"
                               % (L, L, tuple(cr_latent.shape), tuple(cr_face.shape)))
        k = e.cond_key
        vf, vl = _version(cr_face), _version(cr_latent)      # None under torch.inference_mode(): never a hit
        if (self.cache_conditioning and k is not None and vf is not None and vl is not None and k[0] is cr_face and k[1] == vf
                and k[2] is cr_latent and k[3] == vl and e.batch == B):
            return
        e.cond_key = None
        e.prepare(cr_latent, cr_face=cr_face)
        # strong references: identity + version, never addresses (a freed tensor's address is handed to the next batch)
        e.cond_key = (cr_face, vf, cr_latent, vl) if (self.cache_conditioning and vf is not None and vl is not None) else None

    def invalidate_conditioning(self):
        """Forget the cached conditioning (after writing into cr_face / cr_latent through `.data`, numpy or a DLPack alias,
        which does not bump the version counter the cache keys on)."""
        self._engine.cond_key = None
        self._engine.prior_key = None

    def forward(self, latents, timesteps, cr_face, cr_latent):
        if latents.shape[0] == 0:                          # empty batch: like the reference's convs, an empty result
            return UNet2DOutput(torch.empty_like(latents, dtype=torch.float32))
        self.prepare(cr_face, cr_latent)
        return UNet2DOutput(self._engine.eps(latents, timesteps))
