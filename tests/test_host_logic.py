"""CPU tests of the host side: C-ABI surface, parameter manifest, synthetic generator, scheduler tables,
batch sharding over torch.distributed (gloo, world_size 2).  No GPU compute is called."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from hifidiff_amd import _lib, arch, distributed, schedulers, synth
from oracle import hifidiff_oracle as O


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hifidiff_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)                 # comments mention hd_schedule(...) etc.
    declared = set(re.findall(r"\b(hd_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hifidiff_amd.refiner import FacialRefiner
    m = FacialRefiner(16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 16, 16), 0, torch.zeros(1, 3, 128, 128), torch.zeros(1, 4, 16, 16))
    ctx = ctypes.c_void_p()
    rc = _lib.lib().hd_create(ctypes.byref(ctx), 16, 0)
    assert rc < 0 and b"HIP" in _lib.lib().hd_last_error(None) or b"device" in _lib.lib().hd_last_error(None)


def test_manifest_counts():
    man = arch.refiner_manifest(16)
    assert len(man) == 1460
    n = sum(int(np.prod(s)) if len(s) else 1 for s, _, _ in man.values())
    assert n == 563_541_665
    assert arch.refiner_manifest(32)["denoiser.idc_conv.weight"][0] == (8192, 2048, 1, 1)
    assert man["denoiser.middle_blks.3.conv1.weight"][0] == (4096, 2048, 1, 1)


def test_synth_is_deterministic_and_scaled():
    a = synth.make_tensor("denoiser.intro.weight", (128, 4, 3, 3), "conv_w", 36)
    b = synth.make_tensor("denoiser.intro.weight", (128, 4, 3, 3), "conv_w", 36)
    assert np.array_equal(a, b) and abs(a).max() <= 1 / 6 + 1e-7
    g = synth.make_tensor("x.beta", (1, 64, 1, 1), "res_scale", 0)
    assert g.std() > 0.05                                              # beta/gamma are not zero: blocks are not identities
    x, crl, crf = synth.sample_inputs(3, 16)
    x2, _, _ = synth.sample_inputs(2, 16)
    assert torch.equal(x[:2], x2) and crf.min() >= 0 and crf.max() < 1  # per-face generation: shards see the same faces
    assert synth.make_state_dict(arch.idc_manifest())["idc.batch_norm1.num_batches_tracked"].shape == ()


@pytest.mark.parametrize("kind", ["ddim", "ddpm"])
def test_scheduler_tables_match_oracle(kind):
    if kind == "ddim":
        s, o = schedulers.DDIMScheduler(clip_sample_range=3.0), O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
        s.set_timesteps(50); o.set_timesteps(50)
    else:
        s, o = schedulers.DDPMScheduler(clip_sample_range=3.0), O.DDPMScheduler(clip_sample=True, clip_sample_range=3.0)
    ts, coef = s.coefficient_table()
    assert [int(t) for t in ts] == o.timesteps
    assert torch.equal(coef, O.step_coefficients(o, kind))
    # the table is kept per schedule (12 ms of host scalar arithmetic at 1000 steps): the same tensors until the schedule changes
    ts2, coef2 = s.coefficient_table()
    assert ts2 is ts and coef2 is coef
    s.timesteps = s.timesteps[:7]; o.timesteps = o.timesteps[:7]
    ts3, coef3 = s.coefficient_table()
    assert ts3.numel() == 7 and torch.equal(coef3, O.step_coefficients(o, kind)) and torch.equal(coef3, coef[:7])
    s.clip_sample_range = 1.0
    assert float(s.coefficient_table()[1][0, 2]) == 1.0
    s250 = schedulers.DDIMScheduler(); s250.set_timesteps(250)
    assert [int(t) for t in s250.timesteps[:2]] == [996, 992]
    with pytest.raises(NotImplementedError):
        schedulers.DDIMScheduler(beta_schedule="linear")
    noisy = s.add_noise(torch.ones(2, 4, 2, 2), torch.zeros(2, 4, 2, 2), torch.tensor([0, 999]))
    assert torch.allclose(noisy[0], torch.full((4, 2, 2), float(s.alphas_cumprod[0] ** 0.5)))


def test_shard_ranges():
    assert [distributed.shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    assert [distributed.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert distributed.shard_range(0, 0, 2) == (0, 0)                  # empty batch


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from hifidiff_amd import distributed, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = int(sys.argv[2])
x, _, _ = synth.sample_inputs(n, 16)
mine = distributed.shard(x, rank, world) * 2.0 + 1.0          # stand-in for the per-rank sampling result
full = distributed.gather_faces(mine, n)
assert full.shape == x.shape and torch.equal(full, x * 2.0 + 1.0), "gather mismatch"
t = torch.tensor([float(rank)]); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == world - 1
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("n_faces", [4, 5])
def test_two_rank_shard_and_gather_gloo(tmp_path, n_faces):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + n_faces), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(n_faces)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` (the plain command the driver runs) starts two fresh workers BEFORE any GPU call:
    without a GPU each worker reaches the "needs an MI355X" exit with RANK / WORLD_SIZE / MASTER_* set, and the
    launcher returns non-zero because workers failed (reference: accelerator.prepare, test_refiner.py:173-174)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present: the workers would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    for rank in (0, 1):
        assert re.search(r"bench\.py rank %d/2 \(local %d, master 127\.0\.0\.1:\d+\): needs an MI355X" % (rank, rank), r.stderr), r.stderr
    assert "ranks failed" in r.stderr and r.stdout.strip() == ""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "distributed.gather_faces(" in src and src.index("self_launch(_requested_gpus") < src.index("import torch  # noqa")


def test_bench_launcher_ends_the_job_when_a_rank_dies_early(tmp_path):
    """One rank exits before the rendezvous (HD_BENCH_TEST_FAIL_RANK): the launcher names it, shows its stderr, terminates the
    others and returns non-zero within seconds instead of leaving rank 0 in init_process_group until the driver's timeout.
    The surviving rank is kept alive artificially (a sleeping stand-in for a worker stuck in the rendezvous)."""
    import time
    stub = tmp_path / "bench.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("import torch  # noqa: E402")]
    # the launcher as shipped + a worker body that fails on the marked rank and otherwise hangs like a rank in the rendezvous
    stub.write_text(head + """
import time
if os.environ.get("HD_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK"):
    sys.stderr.write("bench.py rank %s: exiting early (HD_BENCH_TEST_FAIL_RANK)\\n" % os.environ["RANK"])
    raise SystemExit(7)
time.sleep(600)
""")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HD_BENCH_TEST_FAIL_RANK"] = "2"
    t0 = time.time()
    r = subprocess.run([sys.executable, str(stub), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    dt = time.time() - t0
    assert r.returncode != 0 and dt < 30, (r.returncode, dt)
    assert "rank 2 exited with code 7 first" in r.stderr and "exiting early (HD_BENCH_TEST_FAIL_RANK)" in r.stderr, r.stderr
    assert r.stdout.strip() == ""
    # the shipped worker has the same early-exit hook and a finite process-group timeout
    assert 'HD_BENCH_TEST_FAIL_RANK' in src and "timeout=datetime.timedelta" in src


def test_vae_accepts_deprecated_attention_key_names():
    """SD-2.x era VAE checkpoints name the mid-block attention projections query / key / value / proj_attn (as (C, C, 1, 1)
    convs); diffusers 0.32.2 converts them when loading (the reference loads such a file at test_refiner.py:176-178)."""
    from hifidiff_amd import arch
    from hifidiff_amd.vae import AutoencoderKL
    man = arch.vae_manifest()
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    old_sd = {}
    for k, (shape, _, _) in man.items():
        nk, sh = k, tuple(shape)
        for new, old in ren.items():
            if ".attentions.0." + new + "." in k:
                nk = k.replace(".attentions.0." + new + ".", ".attentions.0." + old + ".")
                if k.endswith(".weight"):
                    sh = sh + (1, 1)
        old_sd[nk] = torch.zeros(sh)
    assert any(".query." in k for k in old_sd) and not any(".to_q." in k for k in old_sd)
    old_sd["metadata_entry"] = "not a tensor"
    m = AutoencoderKL()
    res = m.load_state_dict(old_sd)
    assert not res.missing_keys and not res.unexpected_keys
    assert set(m.state_dict()) == set(man)
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in old_sd.items() if not k.endswith("proj_attn.bias")})


def test_cr_manifest_and_synthetic_weights():
    """CoarseRestoration manifest (checked key-for-key against the reference in oracle/make_golden.py's container run):
    664 tensors, nine STN heads whose synthetic last Linear leans to the identity transform."""
    from hifidiff_amd import arch, synth
    man = arch.cr_manifest()
    assert len(man) == 664 and list(man)[0] == "intro.weight" and list(man)[-1] == "decoders.3.sampling.0.weight"
    assert man["encoders.0.stn.fc_loc.0.weight"][0] == (85, 7290) and man["middle_blocks.stn.localization.0.weight"][0] == (8, 512, 3, 3)
    assert [s[1:3] for s in arch.cr_stages()] == [(32, 128), (64, 64), (128, 32), (256, 16), (512, 8), (512, 8), (256, 16), (128, 32), (64, 64)]
    sd = synth.cr_state_dict()
    b = sd["decoders.2.stn.fc_loc.2.bias"]
    assert abs(float(b[0]) - 1.0) < 0.1 and abs(float(b[4]) - 1.0) < 0.1 and float(b[[1, 2, 3, 5]].abs().max()) < 0.1


def test_incremental_build_knows_what_each_unit_includes():
    """_lib.build() recompiles a translation unit when a header it includes (directly or not) is newer than its object file: the stage
    kernels live in hd_stages.hip (hd_face.hpp / hd_xcd.hpp / hd_xcd2.hpp), the host side in hd_lib.hip + hd_aux.hip over hd_internal.hpp,
    and none of the GEMM launch tables depends on a stage kernel (an edit there must not rebuild them: 2 minutes each)."""
    import os
    deps = {os.path.basename(u): {os.path.basename(d) for d in _lib.unit_deps(u)} for u in _lib.UNITS}
    assert {"hd_face.hpp", "hd_xcd.hpp", "hd_xcd2.hpp", "hd_stage_api.hpp", "hd_chain.hpp", "hd_gemm.hpp"} <= deps["hd_stages.hip"]
    assert {"hd_internal.hpp", "hd_stage_api.hpp", "hd_gemm.hpp", "hifidiff_hip.h"} <= deps["hd_lib.hip"] and "hd_xcd2.hpp" not in deps["hd_lib.hip"]
    assert "hd_internal.hpp" in deps["hd_aux.hip"]
    for u in ("hd_dispatch_ln.hip", "hd_dispatch_lnface.hip", "hd_dispatch_bf16.hip", "hd_dispatch_misc.hip"):
        assert deps[u] == {u, "hd_dispatch.hpp", "hd_gemm.hpp", "hd_wide.hpp"}, deps[u]      # hd_wide.hpp: a GEMM kernel of the launch table, not a stage
    assert {"hd_strip.hpp", "hd_chain.hpp", "hd_stage_api.hpp"} <= deps["hd_strip.hip"] and "hd_strip.hpp" not in deps["hd_lib.hip"]
    assert set(os.path.basename(p) for p in _lib.SOURCES) >= set().union(*deps.values()) - {"hifidiff_hip.h"}


def test_scheduler_table_cache_follows_everything_coef_reads():
    """the coefficient table depends on alphas_cumprod, final_alpha_cumprod and num_train_timesteps too; a stale table must
    not be served after any of them changes, and plain iterables are accepted as timesteps again."""
    s = schedulers.DDIMScheduler(clip_sample_range=3.0)
    s.set_timesteps(50)
    _, c0 = s.coefficient_table()
    s.alphas_cumprod.mul_(0.5)                                          # in place: same tensor object, new version
    _, c1 = s.coefficient_table()
    assert c1 is not c0 and not torch.equal(c1, c0)
    s.final_alpha_cumprod = torch.tensor(0.9)
    _, c2 = s.coefficient_table()
    assert c2 is not c1 and not torch.equal(c2[-1], c1[-1])             # the last step reads final_alpha_cumprod
    s.timesteps = [int(t) for t in s.timesteps[:5]]                     # a list, not a tensor
    ts, c3 = s.coefficient_table()
    assert ts.numel() == 5 and torch.equal(c3, c2[:5])


def test_bench_latency_floor_adds_up_its_terms():
    """roofline.latency_floor_ms = dependent launches x boundary + in-launch hand-offs x hand-off + weights / stream rate
    (or the MFMA time at peak if longer), every term with the profiles/ file it was measured in."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)                                          # no GPU call at import
    r = b.latency_floor(16, 63, b.SURVEY_WEIGHT_BYTES[16], b.SURVEY_FLOPS_PER_FACE_STEP[16] * 64)
    t = r["latency_floor_terms_us"]
    assert t["inlaunch_handoffs"] == 92 and abs(t["launch_boundaries"] - 63 * 2.21) < 0.06 and abs(t["handoffs"] - 92 * 0.8) < 0.06
    assert abs(t["weight_stream_or_mfma"] - 722.66e6 / 6.5e12 * 1e6) < 0.06
    assert abs(r["latency_floor_ms"] - (63 * 2.21 + 92 * 0.8 + 111.18) * 1e-3) < 1e-3
    assert b.inlaunch_handoffs(16, 151) == 0 and b.inlaunch_handoffs(32, 151) == 0
    r32 = b.latency_floor(32, 151, b.SURVEY_WEIGHT_BYTES[32], b.SURVEY_FLOPS_PER_FACE_STEP[32] * 64)
    assert abs(r32["latency_floor_terms_us"]["weight_stream_or_mfma"] - 8.2917e9 * 64 / 2.5e15 * 1e6) < 0.06      # MFMA time at peak is the longer one
    for src in r["latency_floor_sources"].values():
        assert os.path.exists(os.path.join(ROOT, src[1])), src


def test_synthetic_weights_can_be_reused_across_latent_sides():
    """bench.py's `secondary` (latent 32) takes every tensor whose spec does not depend on the latent side from the headline's state dict:
    a tensor is a pure function of (name, shape, kind, fan_in, seed), so only idc_conv is generated again and the values are the same."""
    m16, m32 = arch.refiner_manifest(16), arch.refiner_manifest(32)
    differ = sorted(n for n in m32 if m16.get(n) != m32[n])
    assert differ == ["denoiser.idc_conv.bias", "denoiser.idc_conv.weight"], differ
    small = {n: m16[n] for n in list(m16)[:4] + ["denoiser.idc_conv.bias"]}
    P16 = synth.make_state_dict(small)
    P32 = synth.make_state_dict({n: m32[n] for n in small}, reuse=(P16, small))
    for n in small:
        fresh = torch.from_numpy(synth.make_tensor(n, *m32[n])).reshape(tuple(m32[n][0]))
        assert torch.equal(P32[n], fresh), n
        assert (P32[n] is P16[n]) == (m16[n] == m32[n]), n
