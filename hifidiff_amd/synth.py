"""Deterministic synthetic weights and inputs (SURVEY §8d "Synthetic inputs").

No checkpoints, datasets or network exist here, so every benchmark and parity test runs on
weights regenerated from one 64-bit seed by a counter-based generator: tensor `name`, element
`i` -> splitmix64(fnv1a(name) ^ seed, i).  Only integer arithmetic and exactly-representable
float scalings are used, so the values are bit-identical on every machine (the GPU box
regenerates the 2.25 GB of weights instead of receiving them).

Initial scales (chosen so that parity tests are not vacuous — the reference zero-initialises
beta/gamma, conditional_naf.py:100-101, which would turn every NAF block into the identity):
  conv/linear weight, bias : U(-1/sqrt(fan_in), +1/sqrt(fan_in))   (PyTorch default scale)
  LayerNorm2d weight / bias : 1 + 0.1 n / 0.1 n
  beta, gamma               : 0.2 n
  BatchNorm weight / bias   : 1 + 0.1 n / 0.1 n ; running_mean 0.1 n ; running_var U(0.75, 1.25)
where n is a unit-variance sum of twelve 16-bit uniforms (Irwin-Hall, exact in float32).
"""
import numpy as np

from . import arch

WEIGHT_SEED = 0x48494649      # "HIFI"
INPUT_SEED = 1
NOISE_SEED = 2

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _C1
    z = (z ^ (z >> np.uint64(27))) * _C2
    return z ^ (z >> np.uint64(31))


def _hash(key: int, n: int, stream: int = 0):
    """n 64-bit words: splitmix64 finaliser over counter (i+1)*GOLD + key + stream*C."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = i * _GOLD + np.uint64((key + stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        return _mix(z)


def uniform(key: int, n: int, stream: int = 0):
    """float32 in [0,1): top 24 bits of the hash."""
    z = _hash(key, n, stream)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def normalish(key: int, n: int, stream: int = 0):
    """float32, mean 0 / variance 1: sum of twelve 16-bit uniforms minus 6 (|x| <= 6)."""
    acc = np.zeros(n, dtype=np.int64)
    for s in range(3):
        z = _hash(key, n, stream * 3 + s + 101)
        for sh in (0, 16, 32, 48):
            acc += ((z >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    # each field is U{0..65535}: mean 32767.5, variance (65536^2-1)/12
    return ((acc.astype(np.float64) - 12 * 32767.5) / 65536.0).astype(np.float32)


def make_tensor(name: str, shape, kind: str, fan_in: int, seed: int = WEIGHT_SEED):
    n = int(np.prod(shape)) if len(shape) else 1
    key = fnv1a64(name) ^ seed
    if kind in ("conv_w", "lin_w", "dw_w", "bias"):
        bound = np.float32(1.0 / np.sqrt(np.float64(fan_in)))
        v = (uniform(key, n) * np.float32(2.0) - np.float32(1.0)) * bound
    elif kind in ("ln_w", "bn_w"):
        v = np.float32(1.0) + np.float32(0.1) * normalish(key, n)
    elif kind in ("ln_b", "bn_b", "bn_mean"):
        v = np.float32(0.1) * normalish(key, n)
    elif kind == "res_scale":
        v = np.float32(0.2) * normalish(key, n)
    elif kind == "bn_var":
        v = np.float32(0.75) + np.float32(0.5) * uniform(key, n)
    elif kind == "bn_count":
        return np.zeros((), dtype=np.int64)
    else:
        raise ValueError(kind)
    return v.astype(np.float32).reshape(shape)


def make_state_dict(manifest, seed: int = WEIGHT_SEED, as_torch: bool = True, only=None, reuse=None):
    """name -> tensor for every entry of an arch.*_manifest (optionally a prefix filter).
    reuse = (state dict, its manifest) made with the same seed: entries whose (shape, kind, fan_in) are the same in both manifests are taken
    from it instead of being generated again -- a tensor is a pure function of (name, shape, kind, fan_in, seed), so the values are identical."""
    out = {}
    for name, (shape, kind, fan_in) in manifest.items():
        if only is not None and not name.startswith(only):
            continue
        if reuse is not None and name in reuse[0] and reuse[1].get(name) == (shape, kind, fan_in):
            out[name] = reuse[0][name]
            continue
        a = make_tensor(name, shape, kind, fan_in, seed)
        if as_torch:
            import torch
            a = torch.from_numpy(np.ascontiguousarray(a)).reshape(tuple(shape))   # keeps 0-d tensors 0-d
        out[name] = a
    return out


def refiner_state_dict(latent_res=16, seed: int = WEIGHT_SEED, as_torch=True, reuse=None):
    """reuse = (state dict, latent_res it was made for), same seed and as_torch: only the tensors whose shape depends on the latent side are generated."""
    if reuse is not None:
        reuse = (reuse[0], arch.refiner_manifest(reuse[1]))
    return make_state_dict(arch.refiner_manifest(latent_res), seed, as_torch, reuse=reuse)


def cr_state_dict(seed: int = WEIGHT_SEED, wild: bool = False):
    """Synthetic CoarseRestoration weights (arch.cr_manifest).  The last Linear of every STN is scaled towards the
    identity transform the reference initialises it to (models/cr/stn.py:38-41), so theta stays a mild warp instead
    of sampling mostly outside the image.  wild=True: strong warps instead (scales 0.6-1.4, shear and translation up
    to ~0.4 of the half-width), so that a good part of every grid_sample lands outside the image (zero padding,
    stn.py:43-52) and the bilinear footprints straddle the border."""
    import torch
    sd = make_state_dict(arch.cr_manifest(), seed)
    for k in sd:
        if k.endswith("stn.fc_loc.2.weight"):
            sd[k] = sd[k] * 0.05
        elif k.endswith("stn.fc_loc.2.bias"):
            b = sd[k] / sd[k].abs().max().clamp_min(1e-12)            # in [-1, 1], deterministic per STN
            sd[k] = torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]) + (0.4 * b if wild else 0.02 * sd[k])
    return sd


def vae_state_dict(seed: int = WEIGHT_SEED):
    """Synthetic AutoencoderKL weights (arch.vae_manifest: 83.7 M parameters; the SD-2.1 checkpoint is not reachable offline)."""
    return make_state_dict(arch.vae_manifest(), seed)


# ---------------------------------------------------------------- synthetic inputs
def randn(tag: str, shape, seed: int = INPUT_SEED):
    n = int(np.prod(shape))
    return normalish(fnv1a64(tag) ^ seed, n).reshape(shape)


def rand(tag: str, shape, seed: int = INPUT_SEED):
    n = int(np.prod(shape))
    return uniform(fnv1a64(tag) ^ seed, n).reshape(shape)


def sample_inputs(batch: int, latent_res: int = 16, seed: int = INPUT_SEED, as_torch=True):
    """x_T ~ N(0,1), cr_latent ~ 0.8 N(0,1), cr_face ~ U[0,1]  (SURVEY §8d).

    Generated per face (tag includes the face index) so a rank's shard equals the same
    faces of the global batch.
    """
    L = latent_res
    x = np.stack([randn(f"x_T/{b}", (4, L, L), seed) for b in range(batch)])
    crl = np.stack([np.float32(0.8) * randn(f"cr_latent/{b}", (4, L, L), seed) for b in range(batch)])
    crf = np.stack([rand(f"cr_face/{b}", (3, 128, 128), seed) for b in range(batch)])
    if as_torch:
        import torch
        return torch.from_numpy(x), torch.from_numpy(crl), torch.from_numpy(crf)
    return x, crl, crf


def ddpm_noise(step: int, face: int, latent_res: int = 16, seed: int = NOISE_SEED):
    return randn(f"z/{step}/{face}", (4, latent_res, latent_res), seed)
