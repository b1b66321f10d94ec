"""CPU oracle for the HifiDiff refiner sampling path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (torch-CPU functional ops, fp32) of the
reference algorithm, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
*checker*.  Nothing in the product path (hifidiff_amd/) imports it; the product fails loudly when the
HIP library is missing.

Pinned: the network part is checked against golden vectors produced by importing the reference itself
(oracle/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).
PARITY UNPINNED for the scheduler: diffusers==0.32.2 (requirements.txt:6 of the reference) is a
third-party dependency that is not vendored and not installed here; `DDIMScheduler` / `DDPMScheduler`
below restate its published arithmetic (Song et al. 2021 eq. 12, Ho et al. 2020 eq. 7/11) anchored on
the reference call sites test_refiner.py:85-91,166-171 and train_refiner.py:337-348.

Every function works on a flat `{state_dict key: tensor}` mapping `P` with the reference's key
names (hifidiff_amd/arch.py), NCHW fp32 tensors, eval-mode BatchNorm.

`Prec` selects the arithmetic:
  Prec(False)  exact fp32, op order of the reference   -> compared with the golden vectors
  Prec(True)   same maths with the MFMA operands (weights and the activation tile fed to each
               1x1 / 2x2 / 3x3 GEMM) rounded to bf16 at exactly the points the HIP kernels round
               them (DESIGN.md "Numerics"), BatchNorm folded into the conv before rounding.
               HIP-vs-oracle differences are then accumulation order only.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------- precision
class Prec:
    def __init__(self, emulate_bf16=False):
        self.emulate = bool(emulate_bf16)

    def q(self, x):
        """Round an MFMA operand (activation tile or weight) to bf16 (RNE) in emulation mode."""
        return x.to(torch.bfloat16).to(torch.float32) if self.emulate else x


FP32 = Prec(False)
BF16 = Prec(True)


def _gemm_conv(x, w, b, prec, stride=1, padding=0):
    """A dense conv executed as an MFMA GEMM: operands rounded, fp32 accumulate, fp32 bias."""
    return F.conv2d(prec.q(x), prec.q(w), b, stride=stride, padding=padding)


def _gemm_linear(x, w, b, prec):
    return F.linear(prec.q(x), prec.q(w), b)


def _bn_affine(P, p, eps=1e-5):
    """Eval-mode BatchNorm2d as y = x*s + o (torch.nn.BatchNorm2d defaults, eps 1e-5)."""
    s = P[p + ".weight"] / torch.sqrt(P[p + ".running_var"] + eps)
    o = P[p + ".bias"] - P[p + ".running_mean"] * s
    return s, o


def _conv_bn(x, P, conv, bn, prec, stride=1, padding=0, gemm=True):
    """conv -> BatchNorm(eval).  fp32: in that order (as the reference does); bf16 emulation: BN is
    folded into the weight and bias first, then the folded weight is rounded (what the packer stores).
    gemm=False: the HIP path runs this conv as an fp32 VALU kernel (folded, not rounded)."""
    w = P[conv + ".weight"]
    b = P.get(conv + ".bias")
    if not prec.emulate:
        y = F.conv2d(x, w, b, stride=stride, padding=padding)
        return F.batch_norm(y, P[bn + ".running_mean"], P[bn + ".running_var"],
                            P[bn + ".weight"], P[bn + ".bias"], False, 0.0, 1e-5)
    s, o = _bn_affine(P, bn)
    wf = w * s.view(-1, 1, 1, 1)
    bf = o if b is None else b * s + o
    if not gemm:
        return F.conv2d(x, wf, bf, stride=stride, padding=padding)
    return F.conv2d(prec.q(x), prec.q(wf), bf, stride=stride, padding=padding)


# --------------------------------------------------------------------------------------- primitives
def layernorm2d(x, w, b, eps=1e-6, prec=None):
    """Per-pixel LayerNorm over channels, biased variance (reference utils.py:16-24, eps utils.py:47).
    bf16 emulation: the statistics come from the fp32 values, the value being normalised is the bf16 copy
    of the residual stream that the HIP loader reads."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    xv = prec.q(x) if prec is not None else x
    y = (xv - mu) / (var + eps).sqrt()
    return w.view(1, -1, 1, 1) * y + b.view(1, -1, 1, 1)


def simple_gate(x):
    """Channel j times channel j+C (reference utils.py:57-60); works on (B,2C,H,W) and (B,2C)."""
    a, b = x.chunk(2, dim=1)
    return a * b


def sinusoidal_embedding(t, dim=128):
    """[sin(t f_k), cos(t f_k)], f_k = exp(-k ln(1e4)/(dim/2-1)) (models/denoiser/model.py:22-29)."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    e = t.to(torch.float32)[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_embedding(P, t, prefix="denoiser"):
    """time_mlp: SinusoidalPosEmb -> Linear(128,1024) -> SimpleGate -> Linear(512,512)
    (models/denoiser/model.py:152-157).  Kept fp32 in both precisions (HIP FiLM path is fp32)."""
    e = sinusoidal_embedding(t, 128)
    h = F.linear(e, P[prefix + ".time_mlp.1.weight"], P[prefix + ".time_mlp.1.bias"])
    return F.linear(simple_gate(h), P[prefix + ".time_mlp.3.weight"], P[prefix + ".time_mlp.3.bias"])


def film_vectors(P, p, temb):
    """Block FiLM: SimpleGate -> Linear(256,4C), chunked shift_att, scale_att, shift_ffn, scale_ffn
    (conditional_naf.py:18-22,103-110).  Returns four (B,C,1,1) tensors."""
    v = F.linear(simple_gate(temb), P[p + ".mlp.1.weight"], P[p + ".mlp.1.bias"])
    return [c[:, :, None, None] for c in v.chunk(4, dim=1)]


# --------------------------------------------------------------------------------------- NAF blocks
def _tap(taps, name, x):
    """Record an intermediate (tests localise a HIP mismatch op by op with these)."""
    if taps is not None:
        taps[name] = x


def _naf_body(P, p, inp, film, prec, taps=None):
    """Shared body of ConditionalNAFBlock.forward (conditional_naf.py:108-136) and NAFBlock.forward
    (models/fpg/naf.py:105-126); `film` is None for the plain block."""
    x = layernorm2d(inp, P[p + ".norm1.weight"], P[p + ".norm1.bias"], prec=prec)
    if film is not None:
        x = x * (film[1] + 1) + film[0]
    x = _gemm_conv(x, P[p + ".conv1.weight"], P[p + ".conv1.bias"], prec)
    _tap(taps, p + ".conv1", x)
    c2 = x.shape[1]
    x = F.conv2d(x, P[p + ".conv2.weight"], P[p + ".conv2.bias"], padding=1, groups=c2)  # fp32 VALU
    g = simple_gate(x)
    _tap(taps, p + ".conv2_gate_pool", g)
    pooled = g.mean(dim=(2, 3), keepdim=True)                       # pooled from the unrounded gate
    s = _gemm_conv(pooled, P[p + ".sca.1.weight"], P[p + ".sca.1.bias"], prec)
    _tap(taps, p + ".sca", s)
    x = prec.q(g) * s                                               # G is stored bf16 by the HIP path
    x = _gemm_conv(x, P[p + ".conv3.weight"], P[p + ".conv3.bias"], prec)
    y = inp + x * P[p + ".beta"]
    _tap(taps, p + ".conv3", y)
    x = layernorm2d(y, P[p + ".norm2.weight"], P[p + ".norm2.bias"], prec=prec)
    if film is not None:
        x = x * (film[3] + 1) + film[2]
    x = _gemm_conv(x, P[p + ".conv4.weight"], P[p + ".conv4.bias"], prec)
    x = simple_gate(x)
    _tap(taps, p + ".conv4", x)
    x = _gemm_conv(x, P[p + ".conv5.weight"], P[p + ".conv5.bias"], prec)
    out = y + x * P[p + ".gamma"]
    _tap(taps, p + ".conv5", out)
    return out


def cond_naf_block(P, p, x, temb, prec=FP32, taps=None):
    return _naf_body(P, p, x, film_vectors(P, p, temb), prec, taps)


def naf_block(P, p, x, prec=FP32, taps=None):
    return _naf_body(P, p, x, None, prec, taps)


# --------------------------------------------------------------------------------------- HCA
def hca_gates(P, p, f_g, prec=FP32, taps=None, tap_name=None):
    """Channel gate w_c (B,C,1,1) and spatial gate w_s (B,1,H,W) of HybridCrossAttention
    (models/fpg/hca.py:33-48).  They depend on the prior only, so the HIP path computes them once."""
    B = f_g.shape[0]
    tn = tap_name or p
    pooled = (F.adaptive_avg_pool2d(f_g, 1) + F.adaptive_max_pool2d(f_g, 1)).reshape(B, -1)
    _tap(taps, tn + ".pool", pooled)
    h = torch.relu(_gemm_linear(pooled, P[p + ".channel_mlp.0.weight"], P[p + ".channel_mlp.0.bias"], prec))
    _tap(taps, tn + ".channel_mlp.0", h)
    w_c = torch.sigmoid(_gemm_linear(h, P[p + ".channel_mlp.2.weight"], P[p + ".channel_mlp.2.bias"], prec))
    _tap(taps, tn + ".channel_mlp.2", w_c)
    h = torch.relu(_conv_bn(f_g, P, p + ".spatial_mlp.0", p + ".spatial_mlp.1", prec))
    _tap(taps, tn + ".spatial_mlp.0", h)
    w_s = torch.sigmoid(_conv_bn(h, P, p + ".spatial_mlp.3", p + ".spatial_mlp.4", prec, gemm=False))
    _tap(taps, tn + ".spatial_mlp.3", w_s)
    return w_c.reshape(B, -1, 1, 1), w_s


def hca_apply(P, p, w_c, w_s, f_d, prec=FP32):
    """f_o = ReLU(BN(conv3x3(f_d + w_c f_d + w_s f_d))) (models/fpg/hca.py:28-29,21-23)."""
    if prec.emulate:
        f_o = f_d * (1.0 + w_c + w_s)          # one fused gate multiply in the A-loader
    else:
        f_o = f_d + w_c * f_d + w_s * f_d
    return torch.relu(_conv_bn(f_o, P, p + ".fused_mlp.0", p + ".fused_mlp.1", prec, padding=1))


def hca(P, p, f_g, f_d, prec=FP32):
    w_c, w_s = hca_gates(P, p, f_g, prec)
    return hca_apply(P, p, w_c, w_s, f_d, prec)


# --------------------------------------------------------------------------------------- FPG
def _up_shuffle(x, w, r, prec):
    y = _gemm_conv(x, w, None, prec)
    return F.pixel_shuffle(y, r) if r > 1 else y


def fpg(P, x, prefix="fpg", prec=FP32, taps=None):
    """FacialPriorGuidance.forward (models/fpg/model.py:46-64): five prior maps, coarsest first."""
    p = prefix
    x = F.conv2d(x, P[p + ".intro.weight"], P[p + ".intro.bias"], padding=1)      # fp32 direct conv
    _tap(taps, p + ".intro", x)
    skips = []
    for i, n in enumerate((2, 2, 4, 8)):
        for j in range(n):
            x = naf_block(P, f"{p}.encoders.{i}.{j}", x, prec, taps)
        skips.append(x)
        x = _gemm_conv(x, P[f"{p}.downs.{i}.weight"], P[f"{p}.downs.{i}.bias"], prec, stride=2)
        _tap(taps, f"{p}.downs.{i}", x)
    x = _up_shuffle(x, P[p + ".convs.0.0.weight"], 1, prec)
    _tap(taps, p + ".convs.0", x)
    priors = [x]
    for i, skip in zip(range(1, 5), skips[::-1]):
        x = _up_shuffle(x, P[f"{p}.convs.{i}.0.weight"], 2, prec) + skip
        _tap(taps, f"{p}.convs.{i}", x)
        priors.append(x)
    return priors


# --------------------------------------------------------------------------------------- IDC
def _store(x, prec):
    """ResNet activations live in bf16 between the HIP conv kernels."""
    return prec.q(x)


def resnet50(P, x, prefix="idc", prec=FP32, taps=None):
    """ResNet.forward / Bottleneck.forward (models/idc/model.py:39-55,122-135): (B,3,128,128) ->
    (B,2048,1,1).  Every conv has a bias and a BatchNorm (eval)."""
    p = prefix
    x = _store(torch.relu(_conv_bn(x, P, p + ".conv1", p + ".batch_norm1", prec, stride=2, padding=3)), prec)
    _tap(taps, p + ".conv1", x)
    x = F.max_pool2d(x, 3, 2, 1)
    _tap(taps, p + ".max_pool", x)
    for li, nblk in enumerate((3, 4, 6, 3), start=1):
        for b in range(nblk):
            q = f"{p}.layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            idn = x
            y = _store(torch.relu(_conv_bn(x, P, q + ".conv1", q + ".batch_norm1", prec)), prec)
            _tap(taps, q + ".conv1", y)
            y = _store(torch.relu(_conv_bn(y, P, q + ".conv2", q + ".batch_norm2", prec, stride=stride, padding=1)), prec)
            _tap(taps, q + ".conv2", y)
            y = _conv_bn(y, P, q + ".conv3", q + ".batch_norm3", prec)
            if b == 0:
                idn = _store(_conv_bn(x, P, q + ".i_downsample.0", q + ".i_downsample.1", prec, stride=stride), prec)
                _tap(taps, q + ".i_downsample", idn)
            x = _store(torch.relu(y + idn), prec)
            _tap(taps, q + ".conv3", x)
    x = x.mean(dim=(2, 3), keepdim=True)
    _tap(taps, p + ".avgpool", x)
    return x


# --------------------------------------------------------------------------------------- denoiser
def normalize_timesteps(timesteps, batch):
    """Scalar / 0-d / (1,) / (B,) int or float -> fp32 (B,) (models/denoiser/model.py:218-229)."""
    if isinstance(timesteps, (int, float)):
        return torch.full((batch,), float(timesteps), dtype=torch.float32)
    t = torch.as_tensor(timesteps)
    if t.dim() == 0:
        return torch.full((batch,), float(t), dtype=torch.float32)
    t = t.to(torch.float32)
    if t.shape[0] == 1 and batch > 1:
        t = t.expand(batch)
    return t


class Conditioning:
    """Step-invariant conditioning of one batch: priors, HCA gates, idc term (hoisted out of the loop;
    the reference recomputes them every step, models/refiner.py:33-34)."""

    def __init__(self, P, cr_latent, cr_face=None, id_emb=None, prec=FP32, prefix="denoiser", taps=None):
        self.priors = fpg(P, cr_latent, "fpg", prec, taps)
        if id_emb is None:
            id_emb = resnet50(P, cr_face, "idc", prec, taps)
        self.id_emb = id_emb
        self.gates = [hca_gates(P, f"{prefix}.hcas.{i}", self.priors[i], prec, taps, f"hcas.{i}") for i in range(5)]
        self.idc = _gemm_conv(id_emb, P[prefix + ".idc_conv.weight"], P[prefix + ".idc_conv.bias"], prec)
        _tap(taps, "idc_conv", self.idc)


def fused_denoiser(P, latents, timesteps, priors=None, id_emb=None, prec=FP32, prefix="denoiser", cond=None,
                   taps=None):
    """FusedDenoiser.forward (models/denoiser/model.py:217-266) -> eps (B,4,L,L).

    Either pass `priors` + `id_emb` (the reference signature) or a prepared `Conditioning`."""
    p = prefix
    B = latents.shape[0]
    t = normalize_timesteps(timesteps, B)
    temb = time_embedding(P, t, p)
    if cond is None:
        gates = [hca_gates(P, f"{p}.hcas.{i}", priors[i], prec) for i in range(5)]
        idc = _gemm_conv(id_emb, P[p + ".idc_conv.weight"], P[p + ".idc_conv.bias"], prec)
    else:
        gates, idc = cond.gates, cond.idc
    x = F.conv2d(latents, P[p + ".intro.weight"], P[p + ".intro.bias"], padding=1)
    _tap(taps, "intro", x)
    skips = []
    for i, n in enumerate((2, 2, 4, 8)):
        for j in range(n):
            x = cond_naf_block(P, f"{p}.encoders.{i}.{j}", x, temb, prec, taps)
        skips.append(x)
        x = _gemm_conv(x, P[f"{p}.downs.{i}.weight"], P[f"{p}.downs.{i}.bias"], prec, stride=2)
        _tap(taps, f"downs.{i}", x)
    for j in range(8):
        x = cond_naf_block(P, f"{p}.middle_blks.{j}", x, temb, prec, taps)
    x = x + idc.reshape(B, *x.shape[1:])
    x = hca_apply(P, p + ".hcas.0", gates[0][0], gates[0][1], x, prec)
    _tap(taps, "hcas.0", x)
    for i, skip in enumerate(skips[::-1]):
        x = _up_shuffle(x, P[f"{p}.ups.{i}.0.weight"], 2, prec) + skip
        _tap(taps, f"ups.{i}", x)
        for j in range(2):
            x = cond_naf_block(P, f"{p}.decoders.{i}.{j}", x, temb, prec, taps)
        x = hca_apply(P, f"{p}.hcas.{i + 1}", gates[i + 1][0], gates[i + 1][1], x, prec)
        _tap(taps, f"hcas.{i + 1}", x)
    x = F.conv2d(x, P[p + ".ending.weight"], P[p + ".ending.bias"], padding=1)
    _tap(taps, "ending", x)
    return x


def denoiser_uncond(P, latents, timesteps, prec=FP32, prefix="denoiser", taps=None):
    """Denoiser.forward (models/denoiser/model.py:112-134): the UNet without priors, HCAs and identity term."""
    p = prefix
    B = latents.shape[0]
    t = normalize_timesteps(timesteps, B)
    temb = time_embedding(P, t, p)
    x = F.conv2d(latents, P[p + ".intro.weight"], P[p + ".intro.bias"], padding=1)
    _tap(taps, "intro", x)
    skips = []
    for i, n in enumerate((2, 2, 4, 8)):
        for j in range(n):
            x = cond_naf_block(P, f"{p}.encoders.{i}.{j}", x, temb, prec, taps)
        skips.append(x)
        x = _gemm_conv(x, P[f"{p}.downs.{i}.weight"], P[f"{p}.downs.{i}.bias"], prec, stride=2)
        _tap(taps, f"downs.{i}", x)
    for j in range(8):
        x = cond_naf_block(P, f"{p}.middle_blks.{j}", x, temb, prec, taps)
    for i, skip in enumerate(skips[::-1]):
        x = _up_shuffle(x, P[f"{p}.ups.{i}.0.weight"], 2, prec) + skip
        _tap(taps, f"ups.{i}", x)
        for j in range(2):
            x = cond_naf_block(P, f"{p}.decoders.{i}.{j}", x, temb, prec, taps)
    x = F.conv2d(x, P[p + ".ending.weight"], P[p + ".ending.bias"], padding=1)
    _tap(taps, "ending", x)
    return x


# ------------------------------------------------------------------------- CoarseRestoration (f1)
def stn_theta(P, p, x):
    """Localisation network of STNBlock -> theta (B,2,3) (models/cr/stn.py:22-48).  fp32 throughout: the
    affine parameters steer a resampling grid, so the HIP path keeps this small network out of bf16 as well."""
    xs = F.conv2d(x, P[p + ".localization.0.weight"], P[p + ".localization.0.bias"])
    xs = torch.relu(F.max_pool2d(xs, 2, stride=2))
    xs = F.conv2d(xs, P[p + ".localization.3.weight"], P[p + ".localization.3.bias"])
    xs = torch.relu(F.max_pool2d(xs, 2, stride=2))
    xs = xs.reshape(x.shape[0], -1)
    h = torch.relu(F.linear(xs, P[p + ".fc_loc.0.weight"], P[p + ".fc_loc.0.bias"]))
    return F.linear(h, P[p + ".fc_loc.2.weight"], P[p + ".fc_loc.2.bias"]).reshape(-1, 2, 3)


def stn_block(P, p, x, taps=None):
    """STNBlock.forward: affine_grid + bilinear grid_sample, align_corners=False, zero padding (stn.py:43-52)."""
    theta = stn_theta(P, p, x)
    _tap(taps, p + ".theta", theta)
    grid = F.affine_grid(theta, list(x.shape), align_corners=False)
    y = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    _tap(taps, p, y)
    return y


def coarse_restoration(P, x, prec=FP32, prefix="", taps=None):
    """CoarseRestoration.forward (models/cr/model.py:73-88): (B,3,128,128) -> (B,3,128,128)."""
    from hifidiff_amd import arch
    q = prefix + "." if prefix else ""
    x = F.conv2d(x, P[q + "intro.weight"], P[q + "intro.bias"], padding=1)
    _tap(taps, q + "intro", x)
    skips = []
    for name, c, r, n, samp in arch.cr_stages():
        sp = q + name
        if name.startswith("decoders"):
            x = x + skips.pop()                                   # model.py:82-83: x = x + enc_skip, then the stage
        for j in range(n):
            x = naf_block(P, f"{sp}.nfbs.{j}", x, prec, taps)
        x = stn_block(P, sp + ".stn", x, taps)
        if samp == "down":
            x = _gemm_conv(x, P[sp + ".sampling.weight"], P[sp + ".sampling.bias"], prec, stride=2)
            skips.append(x)                                       # model.py:77-79: the skip is the stage OUTPUT
        elif samp == "up":
            x = _up_shuffle(x, P[sp + ".sampling.0.weight"], 2, prec)
        _tap(taps, sp, x)
    x = F.conv2d(x, P[q + "outro.weight"], P[q + "outro.bias"], padding=1)
    _tap(taps, q + "outro", x)
    return x


def refiner_forward(P, latents, timesteps, cr_face, cr_latent, prec=FP32):
    """FacialRefiner.forward as written: FPG and IDC recomputed on every call (models/refiner.py:32-38)."""
    priors = fpg(P, cr_latent, "fpg", prec)
    id_emb = resnet50(P, cr_face, "idc", prec)
    return fused_denoiser(P, latents, timesteps, priors, id_emb, prec)


# --------------------------------------------------------------------------------------- schedulers
class _SchedulerBase:
    """Noise schedule shared by DDIM/DDPM: scaled_linear betas, fp32 cumprod (diffusers 0.32.2
    semantics; constructor args as at test_refiner.py:166-171 / train_refiner.py:337-348)."""

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02,
                 beta_schedule="scaled_linear", prediction_type="epsilon",
                 clip_sample=True, clip_sample_range=1.0):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon":
            raise NotImplementedError("only scaled_linear / epsilon are used by the reference")
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)          # set_alpha_to_one=True
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.timesteps = list(range(num_train_timesteps - 1, -1, -1))
        self.num_inference_steps = num_train_timesteps

    def set_timesteps(self, n, device=None):
        """'leading' spacing, steps_offset 0: [i * (T // n)] reversed."""
        r = self.num_train_timesteps // n
        self.num_inference_steps = n
        self.timesteps = [i * r for i in range(n)][::-1]

    def _alpha(self, t):
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod


class _StepOut:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler(_SchedulerBase):
    def step(self, eps, t, x, eta=0.0):
        if eta != 0.0:
            raise NotImplementedError("the reference only samples with eta=0")
        t = int(t)
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a, a_p = self._alpha(t), self._alpha(t_prev)
        x0 = (x - (1 - a) ** 0.5 * eps) / a ** 0.5
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        # use_clipped_model_output=False: eps is not recomputed from the clipped x0
        return _StepOut(a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps, x0)


class DDPMScheduler(_SchedulerBase):
    """'fixed_small' variance; noise is passed in so runs are reproducible."""

    def step(self, eps, t, x, noise=None):
        t = int(t)
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a, a_p = self._alpha(t), self._alpha(t_prev)
        alpha_t = a / a_p
        beta_t = 1 - alpha_t
        x0 = (x - (1 - a) ** 0.5 * eps) / a ** 0.5
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        mu = (a_p ** 0.5 * beta_t / (1 - a)) * x0 + (alpha_t ** 0.5 * (1 - a_p) / (1 - a)) * x
        if t > 0:
            var = torch.clamp((1 - a_p) / (1 - a) * beta_t, min=1e-20)
            mu = mu + var ** 0.5 * noise
        return _StepOut(mu, x0)


def step_coefficients(sched, kind):
    """Per-step scalars in the form the HIP sampler consumes:
        x0     = clamp((x - c[0]*eps) / c[1], +-c[2])
        x_prev = c[3]*x0 + c[4]*x + c[5]*eps + c[6]*z
    One row per entry of sched.timesteps (fp32)."""
    rows = []
    r = sched.num_train_timesteps // sched.num_inference_steps
    clip = float(sched.clip_sample_range) if sched.clip_sample else float("inf")
    for t in sched.timesteps:
        a, a_p = sched._alpha(t), sched._alpha(t - r)
        if kind == "ddim":
            rows.append([(1 - a) ** 0.5, a ** 0.5, clip, a_p ** 0.5, 0.0, (1 - a_p) ** 0.5, 0.0])
        else:
            alpha_t = a / a_p
            beta_t = 1 - alpha_t
            sig = torch.clamp((1 - a_p) / (1 - a) * beta_t, min=1e-20) ** 0.5 if t > 0 else 0.0
            rows.append([(1 - a) ** 0.5, a ** 0.5, clip, a_p ** 0.5 * beta_t / (1 - a),
                         alpha_t ** 0.5 * (1 - a_p) / (1 - a), 0.0, sig])
    return torch.tensor([[float(v) for v in row] for row in rows], dtype=torch.float32)


# --------------------------------------------------------------------------------------- samplers
def sample(P, x, cr_face, cr_latent, sched, kind="ddim", noise_fn=None, prec=FP32,
           as_written=False, id_emb=None, max_steps=None):
    """The reverse-diffusion loop (test_refiner.py:85-91; canonical train_refiner.py:109-120) in
    latent space.  as_written=True recomputes FPG+IDC per step exactly like the reference; otherwise
    the conditioning is hoisted.  `noise_fn(step_index)` supplies z for DDPM."""
    cond = None if as_written else Conditioning(P, cr_latent, cr_face, id_emb, prec)
    for i, t in enumerate(sched.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        tb = torch.full((x.shape[0],), t)
        if as_written:
            eps = refiner_forward(P, x, tb, cr_face, cr_latent, prec)
        else:
            eps = fused_denoiser(P, x, tb, prec=prec, cond=cond)
        if kind == "ddim":
            x = sched.step(eps, t, x, eta=0.0).prev_sample
        else:
            z = noise_fn(i) if t > 0 else None
            x = sched.step(eps, t, x, noise=z).prev_sample
    return x


# --------------------------------------------------------------------------------------- VAE boundary
# AutoencoderKL of "stable-diffusion-2-1-base" as the reference uses it (test_refiner.py:78-83,93,176-178).  THIRD PARTY:
# diffusers==0.32.2 (requirements.txt:6) is neither vendored nor installed and the checkpoint is not reachable, so this is a
# restatement of its published architecture (AutoencoderKL / Encoder / Decoder / ResnetBlock2D / Attention, heads = 1,
# GroupNorm(32, eps 1e-6), Downsample2D pad (0,1,0,1) + stride-2 conv, Upsample2D nearest 2x + conv): PARITY UNPINNED --
# no reference fixture exists for it; the tests pin the HIP path against this restatement only.
def _vae_gn(x, P, p, silu):
    y = F.group_norm(x, 32, P[p + ".weight"], P[p + ".bias"], eps=1e-6)
    return F.silu(y) if silu else y


def vae_resnet(P, p, x, prec=FP32):
    h = _gemm_conv(_vae_gn(x, P, p + ".norm1", True), P[p + ".conv1.weight"], P[p + ".conv1.bias"], prec, padding=1)
    h = _gemm_conv(_vae_gn(h, P, p + ".norm2", True), P[p + ".conv2.weight"], P[p + ".conv2.bias"], prec, padding=1)
    if (p + ".conv_shortcut.weight") in P:
        x = _gemm_conv(x, P[p + ".conv_shortcut.weight"], P[p + ".conv_shortcut.bias"], prec)
    return x + h


def vae_attention(P, p, x, prec=FP32):
    B, C, H, W = x.shape
    h = _vae_gn(x, P, p + ".group_norm", False).reshape(B, C, H * W).transpose(1, 2)
    q = _gemm_linear(h, P[p + ".to_q.weight"], P[p + ".to_q.bias"], prec)
    k = _gemm_linear(h, P[p + ".to_k.weight"], P[p + ".to_k.bias"], prec)
    v = _gemm_linear(h, P[p + ".to_v.weight"], P[p + ".to_v.bias"], prec)
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v          # fp32 in the HIP kernel too
    o = _gemm_linear(a, P[p + ".to_out.0.weight"], P[p + ".to_out.0.bias"], prec)
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def vae_encode_moments(P, x, prec=FP32):
    """AutoencoderKL.encode(x).latent_dist.parameters: [B,8,L,L] (mean | logvar)."""
    h = _gemm_conv(x, P["encoder.conv_in.weight"], P["encoder.conv_in.bias"], prec, padding=1)
    for i in range(4):
        for j in range(2):
            h = vae_resnet(P, f"encoder.down_blocks.{i}.resnets.{j}", h, prec)
        if i < 3:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = _gemm_conv(F.pad(h, (0, 1, 0, 1)), P[p + ".weight"], P[p + ".bias"], prec, stride=2)
    h = vae_resnet(P, "encoder.mid_block.resnets.0", h, prec)
    h = vae_attention(P, "encoder.mid_block.attentions.0", h, prec)
    h = vae_resnet(P, "encoder.mid_block.resnets.1", h, prec)
    h = _gemm_conv(_vae_gn(h, P, "encoder.conv_norm_out", True), P["encoder.conv_out.weight"], P["encoder.conv_out.bias"], prec, padding=1)
    return F.conv2d(h, P["quant_conv.weight"], P["quant_conv.bias"])              # fp32 in the HIP path


def vae_sample(moments, noise):
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def vae_encode_scaled(P, images, image_res, noise, vae_range=False, prec=FP32):
    """cr_latent of test_refiner.py:78-83 (vae_range: train_refiner.py:72-83 applies to_vae_range after the resize)."""
    x = images if images.shape[-1] == image_res else F.interpolate(images, size=(image_res, image_res), mode="bicubic", align_corners=False)
    if vae_range:
        x = x.clamp(0, 1) * 2.0 - 1.0
    return vae_sample(vae_encode_moments(P, x, prec), noise) * 0.18215


def vae_decode_scaled(P, latents, prec=FP32):
    """vae.decode(latents / 0.18215).sample (test_refiner.py:93)."""
    z = F.conv2d(latents / 0.18215, P["post_quant_conv.weight"], P["post_quant_conv.bias"])
    h = _gemm_conv(z, P["decoder.conv_in.weight"], P["decoder.conv_in.bias"], prec, padding=1)
    h = vae_resnet(P, "decoder.mid_block.resnets.0", h, prec)
    h = vae_attention(P, "decoder.mid_block.attentions.0", h, prec)
    h = vae_resnet(P, "decoder.mid_block.resnets.1", h, prec)
    for i in range(4):
        for j in range(3):
            h = vae_resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", h, prec)
        if i < 3:
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            h = _gemm_conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), P[p + ".weight"], P[p + ".bias"], prec, padding=1)
    return _gemm_conv(_vae_gn(h, P, "decoder.conv_norm_out", True), P["decoder.conv_out.weight"], P["decoder.conv_out.bias"], prec, padding=1)
