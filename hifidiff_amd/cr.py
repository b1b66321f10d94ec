"""Drop-in mirror of `CoarseRestoration` (models/cr/model.py:33-88) — the network that turns the low-quality face
into `cr_face` before the diffusion loop (`cr_face = cr_module(ln_face)`, test_refiner.py:77; infer_cr.py:48-60).
SURVEY §8 f1.

Same constructor (no arguments), `load_state_dict` with the reference's keys (`torch.load(ckpt)["model_state_dict"]`,
test_refiner.py:180), `.eval()`, `.to(device)`, `forward(x) -> tensor`.  All compute runs in libhifidiff_hip.so
(`hd_cr_create` / `hd_cr_forward`): the 32 NAF blocks, down- and up-convs on the refiner path's kernels, the STN
(localisation net in fp32, affine bilinear resampling) and the 3<->32 channel image convs on the kernels of
csrc/hd_cr.hpp.  There is no CPU path.
"""
import ctypes

import torch
from torch import nn

from . import _lib, arch


class CoarseRestoration(nn.Module):
    def __init__(self):
        super().__init__()
        self._ctx = None
        self._device = None
        self._state = None
        self._loaded = False
        self._batch = None

    # ---- nn.Module plumbing ----
    def load_state_dict(self, state_dict, strict=True):
        man = arch.cr_manifest()
        missing = [k for k in man if k not in state_dict]
        unexpected = [k for k in state_dict if k not in man]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for CoarseRestoration: Missing key(s): %s; Unexpected key(s): %s"
                               % (missing[:4], unexpected[:4]))
        for k, (shape, _, _) in man.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(shape):
                raise RuntimeError("size mismatch for %s: got %s, expected %s" % (k, tuple(state_dict[k].shape), tuple(shape)))
        base = self._state or {}
        self._state = {k: (state_dict[k] if k in state_dict else base[k]).detach() for k in man if (k in state_dict or k in base)}
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

    # ---- library plumbing ----
    def _create(self, idx):
        ctx = ctypes.c_void_p()
        _lib.check(_lib.lib().hd_cr_create(ctypes.byref(ctx), idx))
        return ctx

    def _ensure(self, device):
        if device.type != "cuda":
            raise RuntimeError("hifidiff_amd runs on an MI355X (gfx950) GPU only; got device %s (no CPU fallback)" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._ctx is not None:
            if idx != self._device.index:
                raise RuntimeError("this model already lives on cuda:%d" % self._device.index)
            return
        self._ctx, self._device = self._create(idx), torch.device("cuda", idx)
        if self._state is not None:
            self._upload()

    def _upload(self):
        man = arch.cr_manifest()
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
        if self._loaded:                     # re-load: weights are packed once per context
            L.hd_destroy(self._ctx)
            self._ctx = None
            self._loaded, self._batch = False, None          # nothing usable until finalize has succeeded
            self._ctx = self._create(self._device.index)
        with torch.cuda.device(self._device):
            _lib.check(L.hd_load_weights(self._ctx, descs, len(man)), self._ctx)
            _lib.check(L.hd_finalize_weights(self._ctx), self._ctx)
        self._loaded, self._batch = True, None

    def forward(self, x):
        self._ensure(x.device)
        if not self._loaded:
            raise RuntimeError("weights are not loaded: call load_state_dict(...) and move the model to a cuda device")
        B = x.shape[0]
        if tuple(x.shape) != (B, 3, 128, 128):
            raise RuntimeError("CoarseRestoration input must be (B,3,128,128), got %s" % (tuple(x.shape),))
        if B == 0:
            return torch.empty_like(x, dtype=torch.float32)
        self._batch = B                                        # any batch size: the library keeps a workspace per recent size
        xin = x.to(device=self._device, dtype=torch.float32).contiguous()
        out = torch.empty_like(xin)
        with torch.cuda.device(self._device):
            _lib.check(_lib.lib().hd_cr_forward(self._ctx, B, xin.data_ptr(), out.data_ptr(),
                                                torch.cuda.current_stream(self._device).cuda_stream), self._ctx)
        self._keep = (xin, out)              # the launch program holds these pointers until the next call
        return out

    def __del__(self):
        try:
            if self._ctx is not None:
                _lib.lib().hd_destroy(self._ctx)
        except Exception:
            pass
