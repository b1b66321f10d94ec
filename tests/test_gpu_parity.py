"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI library, against
 * the CPU oracle in bf16-operand emulation mode (same rounding points: differences are accumulation order),
 * the golden vectors produced by the reference itself (fp32; tolerance from SURVEY §8d: single eps rel-L2
   <= 1e-2, 50-step DDIM latent PSNR >= 40 dB on the clip range 6),
 * size-independent properties at the full benchmark batch (determinism, batch independence, chain split).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, psnr, psnr_pp, rel_l2

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    torch.set_grad_enabled(False)
    return torch.device("cuda", 0)


def make_model(weights, latent=16):
    from hifidiff_amd.refiner import FacialRefiner
    m = FacialRefiner(latent)
    m.load_state_dict(weights)
    m.to("cuda:0")
    return m


@pytest.fixture(scope="module")
def model2(gpu, weights16):
    """Model instance used with batch 2 (the batch size of a context is fixed by its first use)."""
    return make_model(weights16)


@pytest.fixture(scope="module")
def inputs2():
    from hifidiff_amd import synth
    return synth.sample_inputs(2, 16)


def test_library_is_the_one_in_tree(gpu):
    from hifidiff_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "hifidiff_amd")
    assert _lib.lib().hd_create is not None


@pytest.fixture(scope="module")
def model2_launches(gpu, weights16):
    """The program with one launch per GEMM at every level (HD_NO_XCD=1 at creation): 151 launches, every one with a tap."""
    os.environ["HD_NO_XCD"] = "1"
    try:
        return make_model(weights16)
    finally:
        del os.environ["HD_NO_XCD"]


def test_op_by_op_against_oracle(gpu, weights16, model2_launches, inputs2):
    """Every launch of the prologue and of one denoiser evaluation vs the oracle tap of the same name."""
    import op_parity
    model2 = model2_launches
    x, crl, crf = inputs2
    # Bound per launch: 1.5 x the emulation's own noise floor at that tap + 2.5e-3.  The floor is the bf16-operand oracle
    # against itself on inputs perturbed by 2e-7: rounding decisions decorrelate once two evaluations differ by a fraction
    # of a bf16 ulp, so ANY two correct evaluations drift apart along the same curve (1.0e-2 at middle_blks.7.conv5, measured
    # on the oracle alone) -- the r01 "middle-level drift" of 1.3e-2 was this, not a kernel difference.  2.5e-3 covers the
    # bf16 storage of G / G2 (the taps are unrounded fp32).  Early launches therefore have to be within 2.5e-3.
    taps = op_parity.oracle_taps(weights16, x, crl, crf, 500.0)
    floor = op_parity.noise_floor(weights16, x, crl, crf, 500.0, taps)
    assert floor["denoiser.encoders.0.0.conv5"] < 1e-4 < floor["denoiser.middle_blks.7.conv5"] < 3e-2
    for which in (1, 0):
        report = []
        worst = op_parity.scan(model2, weights16, x, crl, crf, 500.0, report, which, taps, floor)
        bad = [r for r in report if "size" in r or "nan" in r]
        assert not bad, bad[:5]
        assert not [r for r in report if "<<<<<<" in r], [r for r in report if "<<<<<<" in r][:10]
        assert worst <= 2.5e-2
    # the first block sees no accumulated drift: fp32-ordering noise only (bf16-stored buffers: one bf16 ulp)
    first = [r for r in report if "encoders.0.0." in r or r.split()[1] == "intro"]
    for r in first:
        rel = float(r.split("rel")[1].split()[0])
        assert rel <= 3e-3, r


def test_eps_against_reference_golden(gpu, model2, inputs2):
    x, crl, crf = [t.cuda() for t in inputs2]
    g = golden("refiner_eps_L16.npz")
    for t in (980, 500, 0):
        eps = model2(x, torch.full((2,), t, device="cuda"), crf, crl).sample
        assert eps.shape == (2, 4, 16, 16) and eps.dtype == torch.float32
        assert rel_l2(eps.cpu(), g[f"eps_t{t}"]) <= 1e-2, t
    # per-face timesteps (model.py:218-229) and the scalar form through FusedDenoiser.forward
    assert rel_l2(model2(x, torch.tensor([37, 861]), crf, crl).sample.cpu(), g["eps_tmixed"]) <= 1e-2
    pri = model2.fpg(crl)
    emb = model2.idc(crf)
    assert emb.shape == (2, 2048, 1, 1) and [tuple(p.shape) for p in pri][0] == (2, 2048, 1, 1)
    eps = model2.denoiser(x, 250, pri, emb).sample
    assert rel_l2(eps.cpu(), g["eps_scalar_t250"]) <= 1e-2


def test_fpg_and_idc_against_reference_golden(gpu, weights16):
    from hifidiff_amd import synth
    m = make_model(weights16)
    _, crl, crf = synth.sample_inputs(1, 16)
    g = golden("fpg_priors.npz")
    for i, p in enumerate(m.fpg(crl.cuda())):
        assert rel_l2(p.cpu(), g[f"prior{i}"]) <= 2e-2, i
    assert rel_l2(m.idc(crf.cuda()).cpu(), golden("idc_embedding.npz")["emb"]) <= 1e-2


def test_ddim50_against_reference_golden(gpu, model2, inputs2):
    """The reference's loop body verbatim (eager) and the graph-replayed loop both land on the golden latent."""
    from hifidiff_amd import sampling, schedulers
    x, crl, crf = [t.cuda() for t in inputs2]
    gd = golden("ddim50_L16.npz")["final"]
    sch = schedulers.DDIMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="epsilon",
                                   clip_sample_range=3.0)
    eager = sampling.ddim_sample_eager(model2, x, crf, crl, sch, 50).cpu()
    sch.set_timesteps(50)
    graph = sampling.sample(model2, x, crf, crl, sch).cpu()
    # measured (profiles/r03_parity_report.txt): PSNR 53.5 dB on the golden's peak-to-peak (6.0: 13 % of it sits on the +-3 clip),
    # rel-L2 6.0e-3.  The old bound (40 dB on a fixed range of 6) accepted an RMSE of 0.06 whatever the signal.
    for got in (eager, graph):
        assert psnr_pp(got, gd) >= 47.0 and rel_l2(got, gd) <= 2e-2, (psnr_pp(got, gd), rel_l2(got, gd))
    assert psnr(eager, graph) >= 50.0                      # same kernels, per-face vs shared FiLM rows
    assert float(graph.abs().max()) <= 3.0 + 1e-6          # last DDIM step returns the clipped x0


def test_ddpm20_with_given_noise_against_reference_golden(gpu, model2, inputs2):
    from hifidiff_amd import sampling, schedulers, synth
    x, crl, crf = [t.cuda() for t in inputs2]
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:20]
    noise = T(np.stack([np.stack([synth.ddpm_noise(i, b, 16) for b in range(2)]) for i in range(20)]))
    out = sampling.sample(model2, x, crf, crl, sch, noise=noise).cpu()
    g = golden("ddpm20_L16.npz")["final"]
    assert rel_l2(out, g) <= 1e-3 and psnr_pp(out, g) >= 75.0, (rel_l2(out, g), psnr_pp(out, g))   # measured 2.6e-5 / 107 dB


def _philox_normal(seed, step, elems):
    """numpy restatement of hd_kernels.hpp: Philox4x32-10, counter (elem, step, 0, 0), Box-Muller."""
    c = [elems.astype(np.uint64), np.full_like(elems, step, dtype=np.uint64), np.zeros_like(elems, dtype=np.uint64),
         np.zeros_like(elems, dtype=np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    u1 = ((c[0] >> np.uint64(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    u2 = (c[1] >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)


def test_scheduler_step_and_device_philox(gpu):
    from hifidiff_amd import _lib
    n = 4096
    x = torch.zeros(n, device="cuda")
    e = torch.zeros(n, device="cuda")
    coef = (ctypes.c_float * 7)(0.0, 1.0, 3.0, 0.0, 0.0, 0.0, 1.0)      # x <- z
    rc = _lib.lib().hd_scheduler_step(x.data_ptr(), e.data_ptr(), coef, None, 0x1234567800000009, 7, n,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    z = _philox_normal(0x1234567800000009, 7, np.arange(n))
    assert np.allclose(x.cpu().numpy(), z, atol=2e-5)
    assert abs(float(x.mean())) < 0.1 and 0.9 < float(x.std()) < 1.1
    # DDIM coefficient form reproduces the oracle's step
    from hifidiff_amd import schedulers
    from oracle import hifidiff_oracle as O
    s, o = schedulers.DDIMScheduler(clip_sample_range=3.0), O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
    s.set_timesteps(50); o.set_timesteps(50)
    xx, ee = torch.randn(2, 4, 16, 16), torch.randn(2, 4, 16, 16)
    got = s.step(ee.cuda(), 500, xx.cuda(), eta=0.0).prev_sample.cpu()
    assert torch.allclose(got, o.step(ee, 500, xx).prev_sample, atol=2e-6)


@pytest.mark.parametrize("batch", [1, 3])
def test_ragged_batches_against_oracle(gpu, weights16, batch):
    """Batches that do not fill a 32-row tile / are not a multiple of anything."""
    from hifidiff_amd import synth
    from oracle import hifidiff_oracle as O
    m = make_model(weights16)
    x, crl, crf = synth.sample_inputs(batch, 16)
    eps = m(x.cuda(), torch.full((batch,), 321), crf.cuda(), crl.cuda()).sample.cpu()
    cond = O.Conditioning(weights16, crl, crf, prec=O.BF16)
    ref = O.fused_denoiser(weights16, x, 321, cond=cond, prec=O.BF16)
    assert rel_l2(eps, ref) <= 6e-3
    if batch == 3:                                           # batch independence: face 0 alone gives the same result
        m1 = make_model(weights16)
        e1 = m1(x[:1].cuda(), torch.full((1,), 321), crf[:1].cuda(), crl[:1].cuda()).sample.cpu()
        assert rel_l2(e1, eps[:1]) <= 6e-3


def test_latent32_against_reference_golden(gpu):
    """FacialRefiner(32): idc_conv 2048->8192 (output permutation), mid at 2x2, level 0 at 32x32 (unfused depthwise path)."""
    from hifidiff_amd import synth
    w32 = synth.refiner_state_dict(32)
    m = make_model(w32, 32)
    x, crl, crf = synth.sample_inputs(1, 32)
    eps = m(x.cuda(), torch.full((1,), 500), crf.cuda(), crl.cuda()).sample.cpu()
    assert rel_l2(eps, golden("refiner_eps_L32.npz")["eps_t500"]) <= 1e-2
    # BASELINE configs[3] in miniature: the first 20 steps of the 250-step DDIM schedule, graph-replayed
    from hifidiff_amd import sampling, schedulers
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    sch.set_timesteps(250)
    sch.timesteps = sch.timesteps[:20]
    lat = sampling.sample(m, x.cuda(), crf.cuda(), crl.cuda(), sch).cpu()
    g20 = golden("ddim250_first20_L32.npz")["final"]                      # RMS 0.157: a fixed range of 6 accepted a 38 % error here
    assert rel_l2(lat, g20) <= 3e-2 and psnr_pp(lat, g20) >= 40.0, (rel_l2(lat, g20), psnr_pp(lat, g20))


def test_every_launch_is_reproducible(gpu, weights16, model2_launches):
    """Each launch of the denoiser program, run three times on the same inputs, gives the same bits (fixed reduction orders,
    no float atomics, and no dependence on how loads and LDS returns happen to be timed: the straight-line K loop once
    failed exactly this, in rows 8j+6 / 8j+7 of one tile in one launch out of tens)."""
    import determinism_scan
    m = make_model(weights16)
    n, bad = determinism_scan.scan(64, 16, 3, model=m, verbose=False, per_face=False)   # one timestep for all faces: the sampling loop's kernels
    assert n == 59 and not bad, (n, bad)                                               # levels 0-3 as eight persistent stages; the intro conv, the first down conv and the last up conv are stage entries; the last HCA conv + the ending conv are one launch
    n, bad = determinism_scan.scan(64, 16, 3, model=model2_launches, verbose=False, per_face=False)   # ... and as one launch per GEMM
    assert n == 151 and not bad, (n, bad)
    n, bad = determinism_scan.scan(64, 16, 3, model=model2_launches, verbose=False, per_face=True)    # a timestep per face: the LdF32LNFace kernels
    assert n == 151 and not bad, (n, bad)
    n, bad = determinism_scan.scan(64, 16, 2, model=m, verbose=False, which=1)      # the conditioning prologue (FPG, IDC, gates)
    assert n > 100 and not bad, bad


def test_full_batch_properties(gpu, weights16):
    """Benchmark batch (64 faces): determinism, agreement with the small-batch run, chain split invariance."""
    from hifidiff_amd import sampling, schedulers, synth
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(64, 16)]
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:10]
    m = make_model(weights16)
    a = sampling.sample(m, x, crf, crl, sch, seed=11)
    b = sampling.sample(m, x, crf, crl, sch, seed=11)
    assert torch.equal(a, b)                                   # bitwise reproducible (fixed reduction orders, no float atomics)
    assert bool(torch.isfinite(a).all()) and float((a - x).abs().mean()) > 1e-3
    c = sampling.sample(m, x, crf, crl, sch, seed=12)
    assert not torch.equal(a, c)                               # the device noise depends on the seed
    os.environ["HD_EXPERIMENTS"] = "1"                         # experiment switch: two sub-batches on two streams (default: one chain)
    os.environ["HD_CHAINS"] = "2"
    try:
        m1 = make_model(weights16)
        a1 = sampling.sample(m1, x, crf, crl, sch, seed=11)
    finally:
        del os.environ["HD_CHAINS"], os.environ["HD_EXPERIMENTS"]
    assert psnr(a1.cpu(), a.cpu()) >= 50.0                     # same faces, same noise indices, different tiling only
    m2 = make_model(weights16)                                 # faces 0,1 sampled alone: noise indices differ -> use DDIM
    d = schedulers.DDIMScheduler(clip_sample_range=3.0); d.set_timesteps(50); d.timesteps = d.timesteps[:10]
    full = sampling.sample(m, x, crf, crl, d)
    two = sampling.sample(m2, x[:2], crf[:2], crl[:2], d)
    assert psnr(two.cpu(), full[:2].cpu()) >= 50.0
    # more faces than one launch of the persistent stages takes (B > 64: per-GEMM launches at every level, and from 128 / 256 faces
    # on the deep-prefetch tall GEMMs at levels 2 / 3): eps of 256 faces = eps of the same faces 64 at a time, tiling apart; 250 faces:
    # the last 128-row tile of those GEMMs is partial (M = 4000 / 1000 rows)
    x4, crl4, crf4 = [t.cuda() for t in synth.sample_inputs(256, 16)]
    big = make_model(weights16)
    for nb in (256, 250):
        e_big = big(x4[:nb], 500, crf4[:nb], crl4[:nb]).sample
        for q in range(4):
            sl = slice(64 * q, min(64 * q + 64, nb))
            e_q = m(x4[sl], 500, crf4[sl], crl4[sl]).sample
            assert rel_l2(e_big[sl].cpu(), e_q.cpu()) <= 6e-3, (nb, q, rel_l2(e_big[sl].cpu(), e_q.cpu()))


def test_static_and_runtime_k_loops_agree_bitwise(gpu, weights16):
    """The straight-line K loop (chunk count known at launch, packed LayerNorm transform) is the same arithmetic as the
    run-time loop it replaced: one eps evaluation of 7 faces gives the same bits with the run-time loop forced in a child
    process (HD_EXPERIMENTS=1 HD_NO_STATIC_K=1; the switch is read once per process)."""
    import subprocess
    import hashlib
    from hifidiff_amd import synth
    code = (
        "import hashlib, torch, sys; sys.path.insert(0, %r)\n"
        "from hifidiff_amd import synth; from hifidiff_amd.refiner import FacialRefiner\n"
        "m = FacialRefiner(16); m.load_state_dict(synth.refiner_state_dict(16)); m.to('cuda:0')\n"
        "x, crl, crf = [t.cuda() for t in synth.sample_inputs(7, 16)]\n"
        "e = m(x, torch.tensor([980., 500., 0., 37., 861., 250., 999.]), crf, crl).sample\n"
        "print('EPS', hashlib.sha256(e.cpu().numpy().tobytes()).hexdigest())\n" % ROOT)
    env = dict(os.environ, HD_EXPERIMENTS="1", HD_NO_STATIC_K="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    child = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("EPS ")][0]
    m = make_model(weights16)
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(7, 16)]
    e = m(x, torch.tensor([980., 500., 0., 37., 861., 250., 999.]), crf, crl).sample
    assert hashlib.sha256(e.cpu().numpy().tobytes()).hexdigest() == child


def test_error_behaviour(gpu, weights16, model2, inputs2):
    from hifidiff_amd.refiner import FacialRefiner
    x, crl, crf = [t.cuda() for t in inputs2]
    with pytest.raises(RuntimeError):
        model2(x[:, :3], 0, crf, crl)                              # wrong latent channels
    with pytest.raises(RuntimeError):
        model2(x, torch.zeros(5), crf, crl)                        # timesteps neither (1,) nor (B,)
    with pytest.raises(RuntimeError):
        model2(x, 0, crf[:, :, :64], crl)                          # cr_face must be 128x128
    m = FacialRefiner(16)
    bad = dict(weights16)
    bad.pop("denoiser.middle_blks.3.conv1.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)                                     # strict: missing key
    bad = dict(weights16)
    bad["denoiser.intro.weight"] = torch.zeros(128, 4, 5, 5)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)                                     # size mismatch
    m2 = FacialRefiner(16)
    m2.to("cuda:0")
    with pytest.raises(RuntimeError):
        m2(x, 0, crf, crl)                                         # weights never loaded
    # empty batch: an empty result, as the reference's convolutions would give
    from hifidiff_amd import sampling, schedulers
    assert tuple(model2(x[:0], 0, crf[:0], crl[:0]).sample.shape) == (0, 4, 16, 16)
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0); sch.set_timesteps(5)
    assert tuple(sampling.sample(model2, x[:0], crf[:0], crl[:0], sch).shape) == (0, 4, 16, 16)


def test_unconditional_denoiser_against_reference_golden(gpu, weights16):
    """the pre-training `Denoiser` (no priors / HCAs / identity) through the same kernels:
    eps vs the reference's outputs and the emulating oracle, the reference's loop body (eager) and the graph loop."""
    from hifidiff_amd import sampling, schedulers, synth
    from hifidiff_amd.refiner import Denoiser
    from oracle import hifidiff_oracle as O
    g = golden("denoiser_uncond_L16.npz")
    m = Denoiser(16)
    n = len("denoiser.")
    sd = {k[n:]: v for k, v in weights16.items() if k.startswith("denoiser.") and ".hcas." not in k and ".idc_conv" not in k}
    with pytest.raises(RuntimeError):
        m.load_state_dict({**sd, "hcas.0.fused_mlp.0.weight": torch.zeros(1)})     # strict: unexpected key
    m.load_state_dict(sd)
    m.to("cuda:0")
    x = T(np.stack([synth.randn(f"x_T/{f}", (4, 16, 16)) for f in range(2)]))
    for i in range(3):
        e = m(x.cuda(), T(g[f"t{i}"]).cuda()).sample.cpu()
        assert rel_l2(e, g[f"eps{i}"]) <= 1e-2, i
    assert rel_l2(m(x.cuda(), 321).sample.cpu(), g["eps_scalar_t"]) <= 1e-2
    e = m(x.cuda(), T(g["t1"]).cuda()).sample.cpu()
    assert rel_l2(e, O.denoiser_uncond(weights16, x, T(g["t1"]), prec=O.BF16)) <= 6e-3
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    eager = sampling.ddim_sample_eager_unconditional(m, x.cuda(), sch, 10).cpu()
    sch.set_timesteps(10)
    graph = sampling.sample(m, x.cuda(), None, None, sch).cpu()
    for got in (eager, graph):
        assert psnr_pp(got, g["ddim10"]) >= 45.0 and rel_l2(got, g["ddim10"]) <= 3e-2, (psnr_pp(got, g["ddim10"]), rel_l2(got, g["ddim10"]))
    with pytest.raises(RuntimeError):
        sampling.sample(m, x.cuda(), x.cuda(), x.cuda(), sch)      # no conditioning inputs in this mode


def test_checkpoint_ingest(gpu, weights16, model2, inputs2, tmp_path):
    """a `model.safetensors` as accelerate's save_state writes it (test_refiner.py:162-164) and the
    constructor's idc .pt / denoiser .safetensors pair (models/refiner.py:18-25) load to the same packed weights."""
    from safetensors.torch import load_file, save_file
    from hifidiff_amd.refiner import FacialRefiner
    x, crl, crf = [t.cuda() for t in inputs2]
    want = model2(x, 500, crf, crl).sample.cpu()
    path = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in weights16.items()}, path)
    m = FacialRefiner(latent_res=16)
    m.load_state_dict(load_file(path))
    m.to("cuda:0")
    assert torch.equal(m(x, 500, crf, crl).sample.cpu(), want)
    os.remove(path)
    # constructor form: IDC from a .pt, denoiser (+ the FPG tensors it shares names with) from a .safetensors
    idc_pt, den_st = str(tmp_path / "idc.pt"), str(tmp_path / "denoiser.safetensors")
    torch.save({"model_state_dict": {k[4:]: v for k, v in weights16.items() if k.startswith("idc.")}}, idc_pt)
    save_file({k[len("denoiser."):]: v.contiguous() for k, v in weights16.items() if k.startswith("denoiser.")}, den_st)
    m3 = FacialRefiner(16, idc_ckpt=idc_pt, denoiser_ckpt=den_st)
    sd = m3.state_dict()
    assert all(torch.equal(sd[k], weights16[k]) for k in sd if k.startswith("idc.") or k.startswith("denoiser."))
    # fpg.* came from the denoiser's same-named encoder tensors (refiner.py:23-24, strict=False)
    assert torch.equal(sd["fpg.encoders.0.0.conv1.weight"], weights16["denoiser.encoders.0.0.conv1.weight"])
    with pytest.raises(RuntimeError):
        m3.to("cuda:0")                                             # fpg.convs.* have no source yet: state dict incomplete


def test_coarse_restoration_against_reference_golden(gpu, model2, inputs2):
    """`CoarseRestoration` (32 NAF blocks + 9 STNs, models/cr/model.py:33-88) through the library:
    output vs the reference's own output (fp32 golden) and vs the bf16-emulating oracle; a second batch size."""
    from hifidiff_amd import synth
    from hifidiff_amd.cr import CoarseRestoration
    from oracle import hifidiff_oracle as O
    g = golden("coarse_restoration.npz")
    P = synth.cr_state_dict()
    m = CoarseRestoration()
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in P.items() if k != "outro.bias"})       # strict: missing key
    m.load_state_dict(P)
    m.to("cuda:0")
    x = T(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(2)]))
    out = m(x.cuda()).cpu()
    assert rel_l2(out, g["out"]) <= 1e-2
    assert rel_l2(out, O.coarse_restoration(P, x, prec=O.BF16)) <= 6e-3
    one = m(x[:1].cuda()).cpu()                                                    # faces are independent
    assert psnr(one, out[:1], data_range=1.0) >= 50.0
    with pytest.raises(RuntimeError):
        m(x[:, :, :64].cuda())                                                     # input must be 128x128
    # the pipeline order of test_refiner.py:77-91: cr_face = cr_module(ln_face) feeds the refiner's sampling loop
    from hifidiff_amd import sampling, schedulers
    xT, crl, _ = [t.cuda() for t in inputs2]
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    sch.set_timesteps(50); sch.timesteps = sch.timesteps[:5]
    lat = sampling.sample(model2, xT, m(x.cuda()), crl, sch)
    assert bool(torch.isfinite(lat).all()) and float((lat - xT).abs().mean()) > 1e-3


def test_graph_reuse_across_schedules(gpu, model2, inputs2):
    """The captured step graph is reused across calls; a longer schedule re-allocates the FiLM table (and must
    re-capture), a shorter one afterwards must give the first result again, bit for bit."""
    from hifidiff_amd import sampling, schedulers
    x, crl, crf = [t.cuda() for t in inputs2]

    def run(n_total, n_run):
        sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
        sch.set_timesteps(n_total)
        sch.timesteps = sch.timesteps[:n_run]
        return sampling.sample(model2, x, crf, crl, sch)
    a = run(50, 6)
    b = run(250, 40)                                   # more FiLM rows than before
    c = run(50, 6)
    assert torch.equal(a, c) and not torch.equal(a, b)
    d = schedulers.DDPMScheduler(clip_sample_range=3.0)
    d.timesteps = d.timesteps[:3]
    e1 = sampling.sample(model2, x, crf, crl, d, seed=5)
    e2 = sampling.sample(model2, x, crf, crl, d, seed=5)
    assert torch.equal(e1, e2) and bool(torch.isfinite(e1).all())


def _read_dbg(model, name, n):
    from hifidiff_amd import _lib
    L = _lib.lib()
    L.hd_debug_read.restype = ctypes.c_int64
    have = L.hd_debug_read(model.engine.ctx, name.encode(), None, 0)
    _lib.check(have, model.engine.ctx)
    assert have >= n, (name, have, n)
    buf = np.empty(have, dtype=np.float32)
    _lib.check(L.hd_debug_read(model.engine.ctx, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), have), model.engine.ctx)
    return torch.from_numpy(buf[:n].copy())


def test_time_embedding_and_film_table_against_reference_golden(gpu, weights16):
    """SinusoidalPosEmb + time_mlp (models/denoiser/model.py:22-29,152-157) for t in {0, 1, 500, 999}: the device buffer
    against the reference's own output, and the folded FiLM rows of every block (conditional_naf.py:103-115) against the
    oracle's film_vectors + LayerNorm affine (the oracle reproduces the reference's blocks bit-exactly on CPU)."""
    from hifidiff_amd import arch, synth
    from oracle import hifidiff_oracle as O
    g = golden("time_embedding.npz")
    assert [float(v) for v in g["t"]] == [0.0, 1.0, 500.0, 999.0]
    m = make_model(weights16)
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(4, 16)]
    m(x, T(g["t"]).cuda(), crf, crl)                                   # per-face timesteps -> 4 FiLM rows
    temb = _read_dbg(m, "temb", 4 * 512).reshape(4, 512)
    assert rel_l2(temb, g["temb"]) <= 2e-5 and float((temb - T(g["temb"])).abs().max()) <= 1e-4
    total = 124928                                                     # sum of 4C over the 32 blocks
    film = _read_dbg(m, "film", 4 * total).reshape(4, total)
    names = [f"denoiser.encoders.{l}.{j}" for l, n in enumerate((2, 2, 4, 8)) for j in range(n)]
    names += [f"denoiser.middle_blks.{j}" for j in range(8)] + [f"denoiser.decoders.{l}.{j}" for l in range(4) for j in range(2)]
    tref = O.time_embedding(weights16, T(g["t"]))
    off = 0
    for p in names:
        sh_a, sc_a, sh_f, sc_f = [v.reshape(4, -1) for v in O.film_vectors(weights16, p, tref)]
        C = sh_a.shape[1]
        w1, b1, w2, b2 = [weights16[p + k] for k in (".norm1.weight", ".norm1.bias", ".norm2.weight", ".norm2.bias")]
        want = torch.cat([b1 * (1 + sc_a) + sh_a, w1 * (1 + sc_a), b2 * (1 + sc_f) + sh_f, w2 * (1 + sc_f)], dim=1)
        got = film[:, off:off + 4 * C]
        assert rel_l2(got, want) <= 2e-5, p
        off += 4 * C
    assert off == total


def test_conditioning_cache_keys_on_identity_not_addresses(gpu, weights16):
    """The loop of ddim_sample passes the same (cr_face, cr_latent) objects on every step (cache hit); the next batch's
    tensors are new objects that PyTorch's caching allocator may place at the freed addresses of the previous batch."""
    from hifidiff_amd import synth
    m = make_model(weights16)
    fresh = make_model(weights16)
    fresh.cache_conditioning = False
    xs, crls, crfs = synth.sample_inputs(4, 16)
    x = xs[:2].cuda()
    t = torch.full((2,), 500, device="cuda")

    def batch(lo):                                   # new out-of-place tensors, as cr_module(...) / vae.encode(...) return
        return (crfs[lo:lo + 2].cuda() * 1.0), (crls[lo:lo + 2].cuda() * 1.0)
    crf_a, crl_a = batch(0)
    pa, pl = crf_a.data_ptr(), crl_a.data_ptr()
    e_a = m(x, t, crf_a, crl_a).sample.clone()
    assert torch.equal(m(x, t, crf_a, crl_a).sample, e_a)               # same objects: cached conditioning, same result
    del crf_a, crl_a
    crf_b, crl_b = batch(2)
    reused = (crf_b.data_ptr() == pa, crl_b.data_ptr() == pl)            # typically (True, True); the test must pass either way
    e_b = m(x, t, crf_b, crl_b).sample
    want = fresh(x, t, crf_b, crl_b).sample
    assert torch.equal(e_b, want), reused
    assert rel_l2(e_b.cpu(), e_a.cpu()) > 1e-2                           # another person's priors give another eps
    crf_b.mul_(0.5)                                                      # in-place edit bumps the version counter: recompute
    assert not torch.equal(m(x, t, crf_b, crl_b).sample, want)


def test_one_model_serves_a_ragged_val_loop(gpu, weights16):
    """val_loop of test_refiner.py:98-112 iterates a DataLoader without drop_last (:160): batches 4, 4, 3 through ONE model
    and ONE CoarseRestoration instance; results equal those of dedicated instances."""
    from hifidiff_amd import sampling, schedulers, synth
    from hifidiff_amd.cr import CoarseRestoration
    m = make_model(weights16)
    xs, crls, crfs = synth.sample_inputs(11, 16)
    outs = []
    for lo, hi in ((0, 4), (4, 8), (8, 11)):
        sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
        outs.append(sampling.ddim_sample_eager(m, xs[lo:hi].cuda(), crfs[lo:hi].cuda(), crls[lo:hi].cuda(), sch, 4))
    assert [o.shape[0] for o in outs] == [4, 4, 3] and all(bool(torch.isfinite(o).all()) for o in outs)
    ded = make_model(weights16)
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    want = sampling.ddim_sample_eager(ded, xs[8:11].cuda(), crfs[8:11].cuda(), crls[8:11].cuda(), sch, 4)
    assert torch.equal(outs[2], want)                                   # same kernels, same tiling: bit-identical
    # back to the first batch size: the parked workspace (programs + graph) is reused and gives the first result again
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    again = sampling.ddim_sample_eager(m, xs[0:4].cuda(), crfs[0:4].cuda(), crls[0:4].cuda(), sch, 4)
    assert torch.equal(again, outs[0])
    sch.set_timesteps(4)
    g4 = sampling.sample(m, xs[0:4].cuda(), crfs[0:4].cuda(), crls[0:4].cuda(), sch)
    g3 = sampling.sample(m, xs[8:11].cuda(), crfs[8:11].cuda(), crls[8:11].cuda(), sch)
    g4b = sampling.sample(m, xs[0:4].cuda(), crfs[0:4].cuda(), crls[0:4].cuda(), sch)
    assert torch.equal(g4, g4b) and psnr(g3.cpu(), outs[2].cpu()) >= 50.0
    model_fwd = m(xs[:1].cuda(), 500, crfs[:1].cuda(), crls[:1].cuda()).sample   # model(x[:1]) after a batch-4 call
    assert tuple(model_fwd.shape) == (1, 4, 16, 16)
    cr = CoarseRestoration()
    cr.load_state_dict(synth.cr_state_dict())
    cr.to("cuda:0")
    ln = T(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(3)]))
    a3 = cr(ln.cuda()).clone()
    a2 = cr(ln[:2].cuda()).clone()
    b3 = cr(ln.cuda())
    assert torch.equal(a3, b3) and psnr(a2.cpu(), a3[:2].cpu(), data_range=1.0) >= 50.0


def test_scheduler_step_argument_checks(gpu):
    from hifidiff_amd import schedulers
    x, e = torch.randn(1, 4, 16, 16, device="cuda"), torch.randn(1, 4, 16, 16, device="cuda")
    d = schedulers.DDPMScheduler(clip_sample_range=3.0)
    d.set_timesteps(250)
    with pytest.raises(ValueError):
        d.step(e, 999, x)                                            # 999 is not in the 250-step schedule: no silent index 0
    assert bool(torch.isfinite(d.step(e, 996, x, seed=3).prev_sample).all())
    with pytest.raises(ValueError):
        schedulers.DDIMScheduler(clip_sample_range=3.0).step(e, 980, x)   # set_timesteps was never called (diffusers raises too)


def test_accelerate_prepare_leaves_the_mirror_unwrapped(gpu, weights16):
    """accelerator.prepare(model) (test_refiner.py:173-174) wraps a model in DDP only if it has parameters that require
    grad; the mirror owns its weights inside the library, so prepare() returns it as it is and the loop runs unchanged."""
    accelerate = pytest.importorskip("accelerate")
    m = make_model(weights16)
    assert list(m.parameters()) == []
    acc = accelerate.Accelerator()
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(5)), batch_size=2)
    m2, loader2 = acc.prepare(m, loader)
    assert m2 is m
    assert [int(b[0].numel()) for b in loader2] == [2, 2, 1]


def test_ddpm_tail_and_mid_slices_against_reference_golden(gpu, model2):
    """DDPM below t = 980: the last 20 steps (t = 19..0, the final step adds no noise) and t = 519..500, graph-replayed with
    the committed noise, and the eager loop body (scheduler.step per iteration) for the tail."""
    from hifidiff_amd import sampling, schedulers, synth
    g = golden("ddpm_slices_L16.npz")
    _, crl, crf = [t.cuda() for t in synth.sample_inputs(2, 16)]
    for name, first, scale in (("tail", 980, 0.7), ("mid", 480, 1.0)):
        sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
        sch.timesteps = sch.timesteps[first:first + 20]
        assert [int(v) for v in sch.timesteps] == [int(v) for v in g[name + "_t"]]
        x = T(np.stack([np.float32(scale) * synth.randn(f"x_{name}/{f}", (4, 16, 16)) for f in range(2)])).cuda()
        noise = T(np.stack([np.stack([synth.ddpm_noise(first + i, b, 16) for b in range(2)]) for i in range(20)]))
        out = sampling.sample(model2, x, crf, crl, sch, noise=noise).cpu()
        assert rel_l2(out, g[name]) <= 1e-3 and psnr_pp(out, g[name]) >= 75.0, (name, rel_l2(out, g[name]), psnr_pp(out, g[name]))   # measured <= 6.1e-5 / >= 100 dB
        if name == "tail":
            lat = x
            for i, t in enumerate(sch.timesteps):
                eps = model2(lat, torch.full((2,), int(t), device="cuda"), crf, crl).sample
                lat = sch.step(eps, t, lat, noise=noise[i].cuda()).prev_sample
            assert rel_l2(lat.cpu(), g[name]) <= 1e-3 and psnr(lat.cpu(), out) >= 50.0
            assert float(out.abs().max()) <= 3.0 + 1e-5             # t = 0: x_prev = mu, a blend of the clipped x0 and x


def test_coarse_restoration_strong_warps_against_reference_golden(gpu):
    """stn_grid_sample_kernel with 28 % of the samples outside the image (zero padding, models/cr/stn.py:43-52)."""
    from hifidiff_amd import synth
    from hifidiff_amd.cr import CoarseRestoration
    from oracle import hifidiff_oracle as O
    g = golden("coarse_restoration_wild.npz")
    P = synth.cr_state_dict(wild=True)
    m = CoarseRestoration()
    m.load_state_dict(P)
    m.to("cuda:0")
    x = T(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(2)]))
    out = m(x.cuda()).cpu()
    assert rel_l2(out, g["out"]) <= 1e-2
    assert rel_l2(out, O.coarse_restoration(P, x, prec=O.BF16)) <= 6e-3


def test_vae_boundary_against_oracle(gpu):
    """bicubic resize -> AutoencoderKL.encode -> posterior sample -> x 0.18215 and / 0.18215 -> decode
    (test_refiner.py:78-83,93) through hd_vae_encode / hd_vae_decode against the CPU restatement of diffusers' AutoencoderKL
    (PARITY UNPINNED: diffusers and the SD-2.1 checkpoint are absent; synthetic weights).  Tolerances: bf16 operands against
    the bf16-emulating oracle 1.5e-2 (moments) / 2.5e-2 (images); against the fp32 oracle 2.5e-2 / 4e-2."""
    from hifidiff_amd import synth
    from hifidiff_amd.vae import AutoencoderKL
    from oracle import hifidiff_oracle as O
    P = synth.vae_state_dict()
    vae = AutoencoderKL()
    with pytest.raises(RuntimeError):
        vae.load_state_dict({k: v for k, v in P.items() if k != "quant_conv.bias"})      # strict: missing key
    vae.load_state_dict(P)
    vae.to("cuda:0")
    x = T(np.stack([synth.rand(f"cr_face_vae/{f}", (3, 128, 128)) for f in range(2)]))
    post = vae.encode(x.cuda()).latent_dist                                              # the reference's call form
    m32, m16 = O.vae_encode_moments(P, x), O.vae_encode_moments(P, x, O.BF16)
    assert tuple(post.parameters.shape) == (2, 8, 16, 16)
    assert rel_l2(post.parameters.cpu(), m16) <= 1.5e-2 and rel_l2(post.parameters.cpu(), m32) <= 2.5e-2
    assert tuple(post.sample().shape) == (2, 4, 16, 16) and torch.equal(post.mode(), post.mean)
    nz = T(np.stack([synth.randn(f"vae_noise/{f}", (4, 16, 16)) for f in range(2)]))
    lat = vae.encode_scaled(x.cuda(), 128, noise=nz.cuda())
    assert rel_l2(lat.cpu(), O.vae_encode_scaled(P, x, 128, nz, prec=O.BF16)) <= 1.5e-2
    a, b = vae.encode_scaled(x.cuda(), 128, seed=5), vae.encode_scaled(x.cuda(), 128, seed=5)
    assert torch.equal(a, b) and not torch.equal(a, vae.encode_scaled(x.cuda(), 128, seed=6))   # device Philox noise
    # decode, both call forms
    z = T(np.stack([np.float32(0.8) * synth.randn(f"vae_z/{f}", (4, 16, 16)) for f in range(2)]))
    img = vae.decode_scaled(z.cuda()).cpu()
    assert tuple(img.shape) == (2, 3, 128, 128)
    assert rel_l2(img, O.vae_decode_scaled(P, z, O.BF16)) <= 2.5e-2 and rel_l2(img, O.vae_decode_scaled(P, z)) <= 4e-2
    assert torch.equal(vae.decode(z.cuda() / 0.18215).sample.cpu(), img)
    # 32 -> 256: bicubic upscale of the 128 x 128 CR output (latent 32), train_refiner.py's to_vae_range variant, one face
    lat32 = vae.encode_scaled(x[:1].cuda(), 256, vae_range=True, noise=torch.zeros(1, 4, 32, 32))
    want = O.vae_encode_scaled(P, x[:1], 256, torch.zeros(1, 4, 32, 32), vae_range=True, prec=O.BF16)
    assert tuple(lat32.shape) == (1, 4, 32, 32) and rel_l2(lat32.cpu(), want) <= 1.5e-2
    with pytest.raises(RuntimeError):
        vae.encode(x[:, :2].cuda())                                                      # wrong channel count
    assert tuple(vae.decode_scaled(z[:0].cuda()).shape) == (0, 3, 128, 128)


def _opt(m, key, v):
    from hifidiff_amd import _lib
    _lib.check(_lib.lib().hd_set_option(m.engine.ctx, key.encode(), int(v)), m.engine.ctx)


def test_xcd_stages_match_the_per_gemm_launches_bit_for_bit(gpu, weights16):
    """Levels 2 and 3 as XCD-local persistent launches (hd_xcd.hpp: 8 faces per XCD, flag-line barrier in the XCD's L2) against
    the per-GEMM launches of the same blocks (hd_set_option "xcd" 0): the same arithmetic in the same order, so the SAME BITS --
    eps at the benchmark batch and at ragged batches, the placement-independent hand-off form, 30 graph-replayed DDPM steps.
    Parity of the per-GEMM launches against the oracle / the reference goldens is what the other tests establish."""
    from hifidiff_amd import _lib, sampling, schedulers, synth
    L = _lib.lib()
    m = make_model(weights16)
    _opt(m, "xcd2", 0)                                                  # this test: hd_xcd.hpp at both levels (level 2 runs hd_xcd2.hpp by default)
    for B in (64, 13, 5):
        x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
        _opt(m, "xcd", 1)
        e1 = m(x, 500, crf, crl).sample.clone()
        assert L.hd_get_option(m.engine.ctx, b"xcd") == 1 and L.hd_get_option(m.engine.ctx, b"xcd_stages") == 8    # block tables of the 4 + 4 stages
        assert L.hd_num_ops(m.engine.ctx, 0) == 59                     # 151 launches with one per GEMM: 80 became 4 (levels 2 / 3), 16 + the intro conv + the first down conv + the last up conv became 4 (levels 0 / 1), hcas.4 + ending became 1
        assert torch.equal(m(x, 500, crf, crl).sample, e1)              # reproducible
        _opt(m, "xcd_force_global", 1)
        eg = m(x, 500, crf, crl).sample.clone()
        _opt(m, "xcd_force_global", 0)
        _opt(m, "xcd", 0)
        e0 = m(x, 500, crf, crl).sample.clone()
        assert bool(torch.isfinite(e0).all())
        assert torch.equal(e1, e0), (B, rel_l2(e1.cpu(), e0.cpu()))
        assert torch.equal(eg, e0), B
        tf = (torch.arange(B, device="cuda") * 7 % 1000).float()        # per-face FiLM rows: the stages step aside
        _opt(m, "xcd", 1)
        ef1 = m(x, tf, crf, crl).sample.clone()
        _opt(m, "xcd", 0)
        assert torch.equal(ef1, m(x, tf, crf, crl).sample)
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:30]
    _opt(m, "xcd", 1)
    a = sampling.sample(m, x, crf, crl, sch, seed=3)
    b = sampling.sample(m, x, crf, crl, sch, seed=3)
    _opt(m, "xcd", 0)
    c = sampling.sample(m, x, crf, crl, sch, seed=3)
    assert torch.equal(a, b) and torch.equal(a, c)
    # batch 2: the per-GEMM launches use 16-row tiles there (another LayerNorm merge tree): equal to accumulation order only
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(2, 16)]
    _opt(m, "xcd", 1)
    e1 = m(x, 500, crf, crl).sample.clone()
    _opt(m, "xcd", 0)
    assert rel_l2(e1.cpu(), m(x, 500, crf, crl).sample.cpu()) <= 4e-3
    _opt(m, "xcd", 1)


def test_autonomous_wave_stages_against_launches_and_oracle(gpu, weights16):
    """hd_xcd2.hpp (the form level 2 runs by default; HD_XCD2=3 at creation builds it for level 3 as well): 16x16x32 MFMAs with
    the weights as the A operand, per-wave hand-offs, weights through an LDS ring.  One K chain per 512 channels instead of
    eight K slices, so it agrees with the per-GEMM launches to accumulation order (eps rel-L2 <= 3e-3 at batch 64, 13, 5),
    bit for bit with itself (replays, the placement-independent hand-off form, 30 graph-replayed DDPM steps), and every
    block of its stages with the oracle on the block's own input (conditional_naf.py:108-136) at both levels."""
    import op_forced
    from hifidiff_amd import _lib, sampling, schedulers, synth
    L = _lib.lib()
    os.environ["HD_XCD2"] = "3"
    try:
        m = make_model(weights16)
        x, crl, crf = [t.cuda() for t in synth.sample_inputs(64, 16)]
        m(x, 500, crf, crl)                                              # finalizes the context under the variable
    finally:
        del os.environ["HD_XCD2"]
    assert L.hd_get_option(m.engine.ctx, b"xcd2") == 3
    for B in (64, 13, 5):
        x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
        _opt(m, "xcd2", 1)
        e2 = m(x, 500, crf, crl).sample.clone()
        assert torch.equal(m(x, 500, crf, crl).sample, e2)              # reproducible
        _opt(m, "xcd_force_global", 1)
        assert torch.equal(m(x, 500, crf, crl).sample, e2)              # write-through hand-offs: the same bits
        _opt(m, "xcd_force_global", 0)
        _opt(m, "xcd", 0)
        e0 = m(x, 500, crf, crl).sample.clone()
        _opt(m, "xcd", 1)
        assert bool(torch.isfinite(e2).all()) and rel_l2(e2.cpu(), e0.cpu()) <= 3e-3, (B, rel_l2(e2.cpu(), e0.cpu()))
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:30]
    a = sampling.sample(m, x, crf, crl, sch, seed=3, check=True)
    assert torch.equal(a, sampling.sample(m, x, crf, crl, sch, seed=3, check=True))
    for B in (2, 64):
        xs, cl, cf = synth.sample_inputs(B, 16)
        rep = []
        worst = op_forced.stage_forced_scan(m, weights16, xs, cl, cf, 500.0, rep)
        assert worst["stages"] == 8 and not [r for r in rep if "<<<<<<" in r], [r for r in rep if "<<<<<<" in r][:8]
        assert worst["fp32"] <= 3e-4 and worst["bf16"] <= 3e-3, (B, worst)


def test_parked_graphs_follow_a_moved_coefficient_buffer(gpu, weights16):
    """A longer schedule re-allocates the scheduler-coefficient buffer; workspaces parked for other batch sizes captured the old
    address into their step graphs (ending launch) and must re-capture too.  Sequence from the r02 review: per-face forward
    (FiLM rows >= steps, so the FiLM table does not move and cannot mask the bug), B = 4 and B = 3 with a short schedule, B = 3
    with a longer one, then B = 4 with the longer one."""
    from hifidiff_amd import sampling, schedulers, synth
    xs, crls, crfs = [t.cuda() for t in synth.sample_inputs(8, 16)]
    m = make_model(weights16)
    m(xs, torch.arange(8, device="cuda") * 100.0, crfs, crls)          # 8 FiLM rows

    def run(model, B, n):
        sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
        sch.set_timesteps(50)
        sch.timesteps = sch.timesteps[:n]
        return sampling.sample(model, xs[:B], crfs[:B], crls[:B], sch)
    run(m, 4, 5); run(m, 3, 5); run(m, 3, 7)
    got = run(m, 4, 7)
    fresh = make_model(weights16)
    fresh(xs, torch.arange(8, device="cuda") * 100.0, crfs, crls)
    assert torch.equal(got, run(fresh, 4, 7))


def test_submodule_calls_do_not_leave_a_stale_conditioning_cache(gpu, weights16):
    """`model.fpg(x)` / `model.idc(x)` with another batch size than the prepared one switch the library's workspace: the next
    forward() with the SAME (cached) conditioning tensors must prepare again instead of failing; FusedDenoiser.forward caches the
    gates on prior identity; load_state_dict on a live model replaces the packed weights."""
    from hifidiff_amd import synth
    m = make_model(weights16)
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(3, 16)]
    e0 = m(x, 500, crf, crl).sample.clone()
    m.fpg(crl[:1]); m.idc(crf[:2])                                      # batches 1 and 2: other workspaces
    assert torch.equal(m(x, 500, crf, crl).sample, e0)                   # same tensor objects as the cached ones
    pri, emb = m.fpg(crl), m.idc(crf)
    d0 = m.denoiser(x, 250, pri, emb).sample.clone()
    assert torch.equal(m.denoiser(x, 250, pri, emb).sample, d0)          # cached gates (same objects)
    pri2 = [p.clone() for p in pri]
    assert torch.equal(m.denoiser(x, 250, pri2, emb).sample, d0)         # new objects, same values: recomputed, same result
    pri2[0].mul_(0.5)
    assert not torch.equal(m.denoiser(x, 250, pri2, emb).sample, d0)     # in-place edit: version counter -> recomputed
    assert torch.equal(m(x, 500, crf, crl).sample, e0)                   # and the refiner path prepares again after it
    w2 = dict(weights16)
    w2["denoiser.ending.bias"] = weights16["denoiser.ending.bias"] + 1.0
    m.load_state_dict(w2)                                                # live model: a fresh context behind the same object
    e1 = m(x, 500, crf, crl).sample
    assert torch.allclose(e1, e0 + 1.0, atol=1e-5)


def test_teacher_forced_op_parity(gpu, weights16, model2_launches):
    """Every launch of one denoiser evaluation against the oracle's arithmetic applied to the launch's OWN inputs (the state
    the HIP path has reached just before it, read back through hd_debug_read): no inherited drift, so the bound is sharp --
    fp32 outputs <= 3e-4, bf16-stored outputs <= 3e-3 (measured r03: 7.6e-5 / 4.9e-4 at batch 2, 3.1e-5 / 2.2e-4 at batch 64) --
    at batch 2 and at the benchmark batch, whose tile shapes differ (16-row tiles, W_NT, CPW = 4, the XCD-affine tile map)."""
    import op_forced
    from hifidiff_amd import synth
    for B in (2, 64):
        x, crl, crf = synth.sample_inputs(B, 16)
        rep = []
        worst = op_forced.forced_scan(model2_launches, weights16, x, crl, crf, 500.0, rep)
        assert len(rep) >= 200 and not [r for r in rep if "no rule" in r], [r for r in rep if "no rule" in r][:3]
        assert not [r for r in rep if "<<<<<<" in r], [r for r in rep if "<<<<<<" in r][:8]
        assert worst["fp32"] <= 3e-4 and worst["bf16"] <= 3e-3, (B, worst)


def test_persistent_stages_block_by_block_against_oracle(gpu, weights16):
    """The eight persistent stages of the program the benchmark times (59 launches: hd_face.hpp levels 0 / 1, hd_xcd.hpp levels
    2 / 3), each stopped after b blocks (`face_block_limit` / `xcd_phase_limit`): the oracle's arithmetic for ONE
    ConditionalNAFBlock (conditional_naf.py:108-136, bf16-operand emulation) applied to the residual stream the stage itself had
    reached before the block, against what the stage holds after it (x' itself as an fp32 output, and the block's own
    contribution x' - x, which passes through the bf16-stored gate tiles, under the bf16 bound) -- 24 blocks + the 8 exit copies
    (bf16 / gated HCA input), at batch 2 and at the benchmark batch.  The stages are checked on their own: nothing here refers to the per-GEMM launches.
    Bounds: fp32 outputs <= 3e-4, bf16-stored <= 3e-3."""
    import op_forced
    from hifidiff_amd import synth
    m = make_model(weights16)
    for B in (2, 64):
        x, crl, crf = synth.sample_inputs(B, 16)
        rep = []
        worst = op_forced.stage_forced_scan(m, weights16, x, crl, crf, 500.0, rep)
        assert worst["stages"] == 8 and len(rep) == 60, (worst, len(rep))      # 24 blocks x 2 + 8 exit copies + the three folded producers (intro, downs.0, ups.3) at their stages' entries + the fused hcas.4 / ending launch
        assert not [r for r in rep if "<<<<<<" in r], [r for r in rep if "<<<<<<" in r][:8]
        assert worst["fp32"] <= 3e-4 and worst["bf16"] <= 3e-3, (B, worst)


def test_eps_at_the_benchmark_batch_against_oracle(gpu, weights16):
    """64 faces, the kernel instantiations the benchmark runs (XCD-local stages included), directly against the bf16-emulating
    oracle: t in {999, 500, 0} with one timestep for all faces, and a timestep per face.  Measured r03: 2.1e-3."""
    from hifidiff_amd import synth
    from oracle import hifidiff_oracle as O
    m = make_model(weights16)
    x, crl, crf = synth.sample_inputs(64, 16)
    cond = O.Conditioning(weights16, crl, crf, prec=O.BF16)
    xd, cld, cfd = x.cuda(), crl.cuda(), crf.cuda()
    for t in (999, 500, 0):
        ref = O.fused_denoiser(weights16, x, t, cond=cond, prec=O.BF16)
        e = m(xd, t, cfd, cld).sample.cpu()
        assert rel_l2(e, ref) <= 6e-3, (t, rel_l2(e, ref))
        assert max(rel_l2(e[f], ref[f]) for f in range(64)) <= 8e-3         # no single face carries the error
    tf = (torch.arange(64) * 37 % 1000).float()
    ref = O.fused_denoiser(weights16, x, tf, cond=cond, prec=O.BF16)
    e = m(xd, tf.cuda(), cfd, cld).sample.cpu()
    assert rel_l2(e, ref) <= 6e-3 and max(rel_l2(e[f], ref[f]) for f in range(64)) <= 8e-3


def test_full_trajectories_against_reference_goldens(gpu, weights16, model2, inputs2):
    """BASELINE configs[1] in miniature: ALL 1000 DDPM steps (clip 3.0, committed-seed noise), B = 2, against the reference
    network run through the same loop (oracle/make_golden.py --only ddpm1000): the latent after 100 / 500 / 1000 chained
    bf16-operand steps.  Stated tolerance: rel-L2 <= 1e-2, PSNR >= 55 dB on the golden's peak-to-peak at the end (measured r03:
    1.3e-3 / 66.1 dB; 7.7e-5 / 97.5 dB after 100 steps).  Then image space: decode through the synthetic-weight VAE (f2, parity
    unpinned) against the fp32 oracle's decode of the REFERENCE latent: >= 50 dB (measured 61.1)."""
    from hifidiff_amd import sampling, schedulers, synth
    from hifidiff_amd.vae import AutoencoderKL
    from oracle import hifidiff_oracle as O
    g = golden("ddpm1000_L16.npz")
    x, crl, crf = [t.cuda() for t in inputs2]
    noise = T(np.stack([np.stack([synth.ddpm_noise(i, b, 16) for b in range(2)]) for i in range(1000)]))
    out = None
    for n, lim_rel, lim_db in ((100, 1e-3, 80.0), (1000, 1e-2, 55.0)):
        sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
        sch.timesteps = sch.timesteps[:n]
        out = sampling.sample(model2, x, crf, crl, sch, noise=noise[:n]).cpu()
        ref = g[f"step{n}"]
        assert rel_l2(out, ref) <= lim_rel and psnr_pp(out, ref) >= lim_db, (n, rel_l2(out, ref), psnr_pp(out, ref))
    assert np.array_equal(g["final"], g["step1000"])
    PV = synth.vae_state_dict()
    vae = AutoencoderKL(); vae.load_state_dict(PV); vae.to("cuda:0")
    img = vae.decode_scaled(out.cuda()).cpu()
    img_ref = O.vae_decode_scaled(PV, T(g["final"]))
    assert psnr_pp(img, img_ref) >= 50.0 and rel_l2(img, img_ref) <= 3e-2, (psnr_pp(img, img_ref), rel_l2(img, img_ref))


_L32_REF = {}


def l32_oracle_eps16():
    """eps of the first 16 synthetic faces at latent 32, t = 500, by the bf16-emulating oracle (computed once per session: ~1 min of CPU)."""
    if "ref" not in _L32_REF:
        from hifidiff_amd import synth
        from oracle import hifidiff_oracle as O
        w32 = synth.refiner_state_dict(32)
        x, crl, crf = synth.sample_inputs(64, 32)
        cond = O.Conditioning(w32, crl[:16], crf[:16], prec=O.BF16)
        _L32_REF["ref"] = O.fused_denoiser(w32, x[:16], 500, cond=cond, prec=O.BF16)
    return _L32_REF["ref"]


def test_latent32_at_batch_against_oracle_and_full_ddim250(gpu):
    """BASELINE configs[3] (32 -> 256 px, 250-step DDIM): eps of 16 faces against the bf16-emulating oracle (the M = 256 .. 16384
    row tile rules, unfused depthwise + pool_finish at level 0); the FULL 250-step DDIM of one face against the reference golden
    (oracle/make_golden.py --only ddim250); and at the configuration's own batch 64 (M up to 65536 rows: 64-row tiles, 16-pixel
    intro / ending runs) ten steps that must be reproducible and agree with the same faces sampled in a batch of 16."""
    from hifidiff_amd import sampling, schedulers, synth
    from oracle import hifidiff_oracle as O
    w32 = synth.refiner_state_dict(32)
    m = make_model(w32, 32)
    x, crl, crf = synth.sample_inputs(64, 32)
    ref = l32_oracle_eps16()
    e = m(x[:16].cuda(), 500, crf[:16].cuda(), crl[:16].cuda()).sample.cpu()
    assert rel_l2(e, ref) <= 6e-3 and max(rel_l2(e[f], ref[f]) for f in range(16)) <= 8e-3, rel_l2(e, ref)
    g = golden("ddim250_L32.npz")
    sch = schedulers.DDIMScheduler(clip_sample_range=3.0)
    sch.set_timesteps(250)
    lat = sampling.sample(m, x[:1].cuda(), crf[:1].cuda(), crl[:1].cuda(), sch).cpu()
    assert rel_l2(lat, g["final"]) <= 3e-2 and psnr_pp(lat, g["final"]) >= 40.0, (rel_l2(lat, g["final"]), psnr_pp(lat, g["final"]))
    sch.timesteps = sch.timesteps[:10]
    xd, cfd, cld = x.cuda(), crf.cuda(), crl.cuda()
    a = sampling.sample(m, xd, cfd, cld, sch)
    b = sampling.sample(m, xd, cfd, cld, sch)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    part = sampling.sample(m, xd[:16], cfd[:16], cld[:16], sch)
    assert rel_l2(part.cpu(), a[:16].cpu()) <= 1e-2                  # other tile shapes at M / 4 rows: accumulation order only
    e64 = m(xd, 500, cfd, cld).sample.cpu()
    assert rel_l2(e64[:16], ref) <= 6e-3                             # the batch-64 instantiations against the oracle as well
    # every launch of the latent-32 program against the oracle's arithmetic on the launch's own inputs (tools/op_forced.py): the
    # unfused conv1 -> depthwise -> pool_finish sequence of level 0 and the strip-staged HCA conv on 32 x 32 faces are only here.
    # Measured r03 (profiles/r03_op_forced_L32_B2.txt): fp32 outputs 5.0e-5, bf16-stored outputs 4.9e-4.
    # At batch 64 the scan runs the configuration's own instantiations: the deep-prefetch 128-row GEMMs of levels 2 / 3, 64-row tiles
    # elsewhere (profiles/r03_op_forced_L32_B64.txt: 2.5e-5 / 2.0e-4).
    import op_forced
    for nb in (2, 64):
        rep = []
        worst = op_forced.forced_scan(m, w32, x[:nb], crl[:nb], crf[:nb], 500.0, rep)
        assert len(rep) >= 200 and not [r for r in rep if "no rule" in r], [r for r in rep if "no rule" in r][:3]
        assert not [r for r in rep if "<<<<<<" in r], [r for r in rep if "<<<<<<" in r][:8]
        assert worst["fp32"] <= 3e-4 and worst["bf16"] <= 3e-3, (nb, worst)


def test_layernorm_gemm_launches_are_reproducible_over_300_runs(gpu, weights16, model2_launches):
    """The launch-to-launch difference of r02 (rows 8j+6 / 8j+7 of one tile, >= 1 launch in 60, cause not established: see
    LdF32LN_T::unit_stats) would slip through three repeats half of the time: every launch that contains a LayerNorm GEMM is
    run 300 times on identical inputs at the benchmark batch (shared FiLM row: the sampling loop's kernels), 60 times with a
    timestep per face, and the XCD-local stages (which carry the same transform) 300 times."""
    import determinism_scan
    shallow = ("encoders.0.", "encoders.1.", "decoders.2.", "decoders.3.")            # levels 0 / 1: LN2 + conv4 sit inside the chain kernel ("conv5")
    ln = lambda n: (n.endswith(".conv2_gate_pool") or n.endswith(".conv4") or n.endswith(".conv1") or     # noqa: E731
                    (n.endswith(".conv5") and any(k in n for k in shallow)))
    n, bad = determinism_scan.scan(64, 16, 300, model=model2_launches, verbose=False, per_face=False, only=ln)
    assert n == 64 and not bad, (n, bad)
    n, bad = determinism_scan.scan(64, 16, 60, model=model2_launches, verbose=False, per_face=True, only=ln)
    assert n == 64 and not bad, (n, bad)
    m = make_model(weights16)
    stage = lambda n: n in ("denoiser.encoders.0.1.conv5", "denoiser.encoders.1.1.conv5", "denoiser.encoders.2.3.conv5", "denoiser.encoders.3.7.conv5",   # noqa: E731
                            "denoiser.decoders.0.1.conv5", "denoiser.decoders.1.1.conv5", "denoiser.decoders.2.1.conv5", "denoiser.decoders.3.1.conv5")
    n, bad = determinism_scan.scan(64, 16, 300, model=m, verbose=False, per_face=False, only=stage)
    assert n == 8 and not bad, bad


def test_face_cluster_stages_of_the_shallow_levels(gpu, weights16):
    """Levels 0 / 1 as face-cluster persistent launches (hd_face.hpp: 32 pixel rows per workgroup, the workgroups of a face exchange
    the depthwise halo rows and the pool sums inside the launch) against the per-block launches (fused conv1 + chain kernel,
    hd_set_option "face" 0).  Same arithmetic except the LayerNorm statistics (two-pass in the kernel instead of merged producer
    partials): X after the first stage agrees to 1e-4, eps to 3e-3, and both forms sit at the same distance from the oracle."""
    from hifidiff_amd import _lib, sampling, schedulers, synth
    from oracle import hifidiff_oracle as O
    L = _lib.lib()
    m = make_model(weights16)
    for B in (64, 13, 2):
        x, crl, crf = synth.sample_inputs(B, 16)
        xd, cld, cfd = x.cuda(), crl.cuda(), crf.cuda()
        _opt(m, "face", 1)
        e1 = m(xd, 500, cfd, cld).sample.clone()
        assert L.hd_get_option(m.engine.ctx, b"face_stages") == 4 and L.hd_num_ops(m.engine.ctx, 0) == 59 and L.hd_get_option(m.engine.ctx, b"end_fold") == 1 and L.hd_get_option(m.engine.ctx, b"intro_fold") == 1 and L.hd_get_option(m.engine.ctx, b"down_fold") == 1 and L.hd_get_option(m.engine.ctx, b"up_fold") == 1
        assert torch.equal(m(xd, 500, cfd, cld).sample, e1)
        names = [L.hd_debug_op_name(m.engine.ctx, 0, i).decode() for i in range(59)]
        L.hd_debug_limit_ops(m.engine.ctx, 0, names.index("denoiser.encoders.0.1.conv5") + 1)
        m(xd, 500, cfd, cld)
        xa = _read_dbg(m, "X0", B * 256 * 128)
        _opt(m, "face", 0)
        m(xd, 500, cfd, cld)
        xb = _read_dbg(m, "X0", B * 256 * 128)
        L.hd_debug_limit_ops(m.engine.ctx, 0, -1)
        e0 = m(xd, 500, cfd, cld).sample.clone()
        assert rel_l2(xa, xb) <= 1e-4, (B, rel_l2(xa, xb))
        assert rel_l2(e1.cpu(), e0.cpu()) <= 3e-3, (B, rel_l2(e1.cpu(), e0.cpu()))
        if B <= 16:
            cond = O.Conditioning(weights16, crl, crf, prec=O.BF16)
            ref = O.fused_denoiser(weights16, x, 500, cond=cond, prec=O.BF16)
            assert rel_l2(e1.cpu(), ref) <= 6e-3 and rel_l2(e0.cpu(), ref) <= 6e-3
        tf = (torch.arange(B, device="cuda") * 7 % 1000).float()        # per-face FiLM rows: the stages step aside, bit for bit
        _opt(m, "face", 1)
        ef1 = m(xd, tf, cfd, cld).sample.clone()
        _opt(m, "face", 0)
        assert torch.equal(ef1, m(xd, tf, cfd, cld).sample)
    _opt(m, "face", 1)
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:30]
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(64, 16)]
    a = sampling.sample(m, x, crf, crl, sch, seed=3)
    assert torch.equal(a, sampling.sample(m, x, crf, crl, sch, seed=3))


def test_a_stage_that_gives_up_poisons_its_call_and_reports(gpu, weights16, model2_launches):
    """a hand-off wait of a persistent stage that gives up must not hand back garbage with rc 0.
    Fault injection (hd_set_option "stage_test_abort"): group 0 of the first XCD-local stage gives up its wait for phase 3 /
    face 0 of the first face-cluster stage gives up the pool wait of block 1 / a LOADER wave of the level-2 stage gives up its wait
    for the gain | bias row of phase 3 (a loader used to raise only its workgroup's LDS word, the stage left with partial
    results and hd_check() said OK).  The injected call returns (it only enqueues
    work) and its result is NaN on the device; check() after the synchronisation raises once and names the code; the context
    then runs one launch per GEMM and agrees bit for bit with a model built with HD_NO_XCD=1; the remaining stage launches
    of the failed call step aside at entry (the call does not take n x 0.6 s of spinning)."""
    import time
    from hifidiff_amd import _lib, sampling, schedulers, synth
    L = _lib.lib()
    B = 5
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
    want = model2_launches(x, 500, crf, crl).sample.clone()
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:12]
    want_lat = sampling.sample(model2_launches, x, crf, crl, sch, seed=3).clone()
    for inject, code in ((4, 0x103), (1001, 0x301), (2003, 0x703)):
        m = make_model(weights16)
        good = m(x, 500, crf, crl).sample.clone()
        m.check()                                                       # nothing failed so far
        assert L.hd_get_option(m.engine.ctx, b"xcd") == 1 and bool(torch.isfinite(good).all())
        _opt(m, "stage_test_abort", inject)
        bad = m(x, 500, crf, crl).sample                               # rc 0: the call only enqueues
        torch.cuda.synchronize()
        assert bool(torch.isnan(bad).all()), "the failed call's eps must be NaN, not garbage"
        with pytest.raises(RuntimeError) as ei:
            m.check()
        assert ("0x%x" % code) in str(ei.value) and "stage_test_abort" in str(ei.value), str(ei.value)
        m.check()                                                       # reported once
        _opt(m, "stage_test_abort", 0)
        assert L.hd_get_option(m.engine.ctx, b"xcd") == 0               # fell back
        again = m(x, 500, crf, crl).sample
        m.check()
        assert torch.equal(again, want), rel_l2(again.cpu(), want.cpu())
        # the same inside a graph-replayed loop: the injected loop hands back NaN latents, quickly; the next loop is valid
        m2 = make_model(weights16)
        sampling.sample(m2, x, crf, crl, sch, seed=3, check=True)
        _opt(m2, "stage_test_abort", inject)
        t0 = time.perf_counter()
        lat = sampling.sample(m2, x, crf, crl, sch, seed=3, check=False)   # enqueue only: the default (check=True) raises here
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 20.0                          # one give-up, not one per remaining stage launch
        assert bool(torch.isnan(lat).all())
        with pytest.raises(RuntimeError):
            sampling.sample(m2, x, crf, crl, sch, seed=3)               # the next call reports it on entry
        _opt(m2, "stage_test_abort", 0)
        lat2 = sampling.sample(m2, x, crf, crl, sch, seed=3, check=True)
        assert torch.equal(lat2, want_lat), rel_l2(lat2.cpu(), want_lat.cpu())


def test_inference_mode_tensors_are_served(gpu, model2, inputs2):
    """tensors created under torch.inference_mode() do not track a version counter (reading `_version` raises):
    they must be served (as a cache miss, re-prepared every call), through FacialRefiner.forward and FusedDenoiser.forward."""
    x, crl, crf = inputs2
    want = model2(x.cuda(), 500, crf.cuda(), crl.cuda()).sample.clone()
    with torch.inference_mode():
        xi, cli, cfi = x.cuda(), crl.cuda(), crf.cuda()
        a = model2(xi, 500, cfi, cli).sample.clone()
        b = model2(xi, 500, cfi, cli).sample.clone()
        pri = model2.fpg(cli)
        emb = model2.idc(cfi)
        c = model2.denoiser(xi, 500, pri, emb).sample.clone()
        d = model2.denoiser(xi, 500, pri, emb).sample.clone()
    assert torch.equal(a, want) and torch.equal(b, want)
    assert rel_l2(c.cpu(), want.cpu()) <= 1e-6 and torch.equal(c, d)
    model2.invalidate_conditioning(); model2.denoiser.invalidate_conditioning()
    assert torch.equal(model2(x.cuda(), 500, crf.cuda(), crl.cuda()).sample, want)


def test_multi_rank_code_path_at_one_rank_over_rccl(gpu):
    """the N > 1 path of bench.py (init_process_group("nccl") = RCCL, shard_range, gather_faces through
    all_gather, the timing all_reduce) rehearsed on hardware with ONE rank, as a fresh child process (a process that has
    initialised the GPU never re-execs): rc 0, a world of 1 over RCCL, finite output, the gather went through RCCL."""
    import json
    import subprocess
    env = dict(os.environ, HD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0",
               HD_TRACE_GATHER="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                        "--diffusion-steps", "20", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    cfg = res["config"]
    assert cfg["rccl_world_size"] == 1 and cfg["output_finite"] is True and res["n_gpus"] == 1
    assert cfg["faces_per_rank"] == [64] and cfg["diffusion_steps"] == 20
    assert "gather_faces: all_gather over nccl, world 1" in r.stderr, r.stderr[-2000:]


_L32_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from hifidiff_amd import _lib, synth
from hifidiff_amd.refiner import FacialRefiner
torch.set_grad_enabled(False)
m = FacialRefiner(32); m.load_state_dict(synth.refiner_state_dict(32)); m.to("cuda:0")
x, crl, crf = synth.sample_inputs(64, 32)
e = m(x.cuda(), 500, crf.cuda(), crl.cuda()).sample.cpu()
torch.save({"eps": e, "launches": _lib.lib().hd_num_ops(m.engine.ctx, 0)}, sys.argv[2])
"""


def test_latent32_strip_and_wide_kernels_against_the_launch_forms_they_replace(gpu, tmp_path):
    """hd_strip.hpp (levels 0 / 1) and hd_wide.hpp (levels 2 / 3) at batch 64, latent 32, against the forms they replace (unfused conv1 ->
    depthwise -> pool finish, fused-epilogue GEMM on whole faces, deep-prefetch tall kernels), which a fresh child process selects with
    HD_NO_STRIP / HD_NO_WIDE (read once per process): 151 against 159 launches, the same eps up to accumulation order and the bf16
    rounding of values that differ by it.  Each form on its own is held against the oracle launch by launch in the test above."""
    import subprocess
    from hifidiff_amd import _lib, synth
    out = str(tmp_path / "eps_old_forms.pt")
    env = dict(os.environ, HD_EXPERIMENTS="1", HD_NO_STRIP="1", HD_NO_WIDE="1")
    r = subprocess.run([sys.executable, "-c", _L32_CHILD, ROOT, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    old = torch.load(out)
    m = make_model(synth.refiner_state_dict(32), 32)
    x, crl, crf = synth.sample_inputs(64, 32)
    e = m(x.cuda(), 500, crf.cuda(), crl.cuda()).sample.cpu()
    n = _lib.lib().hd_num_ops(m.engine.ctx, 0)
    assert (n, old["launches"]) == (151, 159), (n, old["launches"])
    assert bool(torch.isfinite(e).all()) and rel_l2(e, old["eps"]) <= 3e-3, rel_l2(e, old["eps"])
    # the same faces twice in a batch of 128 (level 3: 2048 rows, still the wide kernel's 128-row form; level 2: 8192 rows, its 256-row form with
    # 512 workgroups; levels 0 / 1: 1024 and 512 strips): both halves must reproduce the batch-64 result up to tile-rule differences
    m2 = make_model(synth.refiner_state_dict(32), 32)
    x2, crl2, crf2 = torch.cat([x, x]), torch.cat([crl, crl]), torch.cat([crf, crf])
    e2 = m2(x2.cuda(), 500, crf2.cuda(), crl2.cuda()).sample.cpu()
    assert rel_l2(e2[:64], e) <= 3e-3 and rel_l2(e2[64:], e) <= 3e-3, (rel_l2(e2[:64], e), rel_l2(e2[64:], e))
    # ... and against the oracle, not only against themselves: the first 16 faces of both halves of the batch-128 run
    ref = l32_oracle_eps16()
    assert rel_l2(e2[:16], ref) <= 6e-3 and rel_l2(e2[64:80], ref) <= 6e-3, (rel_l2(e2[:16], ref), rel_l2(e2[64:80], ref))
    assert max(rel_l2(e2[f], ref[f]) for f in range(16)) <= 8e-3


def test_the_cost_of_leaving_the_benchmark_shapes_is_bounded(gpu, weights16):
    """the fast paths are pinned to the benchmark's shapes (persistent stages: latent 16, batch <= 64; the wide / deep
    many-row GEMMs of latent 32: its batch-64 row counts).  What leaving them costs is a number (tools/batch_sweep.py,
    profiles/r05_batch_sweep.txt: 1.04 ms per step at batch 64, 1.64 ms at 65, 1.75 ms at 128); asserted here: the launch counts on either
    side of the cliff, that a face at batch 65 costs at most 1.7x the batch-64 face (measured 1.55x) and that the one-launch-per-GEMM
    program is still the better answer there: taking the batch through the persistent stages in passes of 64 would cost two batch-64
    steps (2.08 ms) at any batch from 65 to 128, which is more than the launches take at either end of that range."""
    import batch_sweep
    rows = {r[0]: r for r in batch_sweep.sweep(16, (64, 65, 128), weights=weights16, n_steps=16, reps=2)}
    assert (rows[64][1], rows[65][1], rows[128][1]) == (59, 151, 151), rows
    per_face = {b: rows[b][3] for b in rows}
    assert per_face[65] <= 1.7 * per_face[64], per_face
    assert per_face[128] <= 1.0 * per_face[64], per_face           # measured 0.84x: the per-GEMM launches barely notice the row count
    assert rows[65][2] < 2.0 * rows[64][2] and rows[128][2] < 2.0 * rows[64][2], rows      # passes of 64 would lose at both ends


def test_folded_transitions_against_their_launches(gpu, weights16):
    """r05: the intro conv, downs.0 and ups.3 run as the ENTRY of the face-cluster stage that consumes them and hcas.4 shares a launch with the
    ending conv (59 launches).  HD_NO_INTRO_FOLD=1 at context creation keeps all four as launches of their own (63): the same eps -- the intro
    entry and the fused ending are the launches' arithmetic operation for operation (bit-identical on their own), the two GEMM entries differ
    from their launches by the accumulation order of K and the bf16 roundings that follow -- and the same 24-step DDPM loop to 1e-2."""
    from hifidiff_amd import _lib, sampling, schedulers, synth
    L = _lib.lib()
    m = make_model(weights16)
    os.environ["HD_NO_INTRO_FOLD"] = "1"
    try:
        m0 = make_model(weights16)
    finally:
        del os.environ["HD_NO_INTRO_FOLD"]
    sch = schedulers.DDPMScheduler(clip_sample_range=3.0)
    sch.timesteps = sch.timesteps[:24]
    for B in (64, 3):
        x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
        e1 = m(x, 500, crf, crl).sample.clone()
        e0 = m0(x, 500, crf, crl).sample.clone()
        assert (L.hd_num_ops(m.engine.ctx, 0), L.hd_num_ops(m0.engine.ctx, 0)) == (59, 63)
        assert [L.hd_get_option(m0.engine.ctx, k) for k in (b"intro_fold", b"down_fold", b"up_fold")] == [0, 0, 0]
        assert bool(torch.isfinite(e1).all()) and rel_l2(e1.cpu(), e0.cpu()) <= 3e-3, (B, rel_l2(e1.cpu(), e0.cpu()))
        a = sampling.sample(m, x, crf, crl, sch, seed=5)
        b = sampling.sample(m0, x, crf, crl, sch, seed=5)
        assert rel_l2(a.cpu(), b.cpu()) <= 1e-2, (B, rel_l2(a.cpu(), b.cpu()))
