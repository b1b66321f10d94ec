#!/usr/bin/env python3
"""Headline benchmark: faces/sec, 16->128 px, 1000-step DDPM reverse diffusion, batch 64 per GPU
(BASELINE.json configs[1]; SURVEY §8d "Config 2").

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without RANK/WORLD_SIZE in the environment and N > 1 the parent spawns N fresh worker processes (one per
GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1) BEFORE it imports torch or makes any
GPU call, relays rank 0's JSON line and exits non-zero if any worker fails.  This is the plain-command
form of `accelerator.prepare(val_dataloader, model)` (reference test_refiner.py:173-174).

One "step" = one complete reverse-diffusion pass over the rank's batch of synthetic faces: the
once-per-batch conditioning prologue (FPG, ResNet-50 IDC, HCA gates, idc_conv), the FiLM table for all
timesteps, and `diffusion_steps` graph-replayed denoiser evaluations + scheduler updates, inputs already
resident in HBM.  Faces are independent, so ranks shard the batch with no collective inside the loop;
RCCL is used only for the final gather of the latents (inside the timed region) and the timing reduce.

Prints ONE JSON line (rank 0).  `roofline`: one launch = one captured step graph (one denoiser
evaluation of the whole batch); algorithmic bytes per launch = the bf16 weights every step must stream
(0.7227 GB at latent 16), duration = HIP-event time of the replay loop / diffusion steps.
`roofline.latency_floor_ms` = dependent launches x measured boundary + in-launch hand-offs x measured hand-off + weight bytes / measured chip
stream rate (each term with its profiles/ file); `frac_of_latency_floor` = that / the measured step.  `secondary`: BASELINE configs[3]
(latent 32, batch 64, 250-step DDIM), 3 timed passes after the headline's timed region.
`cpu_baseline`: the CPU oracle (a port of the reference's algorithm, as-written semantics: FPG+IDC
recomputed every step, fp32) timed on this host on a bounded sample and extrapolated linearly.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def self_launch(n):
    """Parent of a plain `python bench.py --gpus N` (N > 1): N child processes, one rank each.  Nothing here
    touches the GPU (torch is not even imported yet): a process that has initialised HIP must never be the
    one that starts or re-execs workers.  All children are polled: when one exits non-zero the others (fresh
    child processes of this parent) are terminated, the failing rank and the tail of its stderr are printed
    and the parent returns non-zero within seconds -- rank 0 is never left waiting in the rendezvous."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="hd_bench_ranks_")
    procs, errs, outs = [], [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HD_BENCH_SELF_LAUNCHED="1")
        errs.append(open(os.path.join(logdir, "rank%d.stderr" % r), "w+"))
        outs.append(open(os.path.join(logdir, "rank%d.stdout" % r), "w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=outs[r], stderr=errs[r]))
    limit = float(os.environ.get("HD_BENCH_RANK_TIMEOUT", "3000"))            # s: a hung job ends with diagnostics, not silently
    t0 = time.time()
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None or time.time() - t0 > limit:
            break
        time.sleep(0.2)
    timed_out = failed is None and any(c is None for c in codes)
    if failed is not None or timed_out:
        grace = time.time() + 3.0                                               # ranks that are failing for the same reason finish by themselves
        while failed is not None and time.time() < grace and any(c is None for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    codes[r] = pr.poll()
            time.sleep(0.1)
        for r, pr in enumerate(procs):                                          # the remaining ranks wait for the dead one: end them
            if codes[r] is None:
                pr.terminate()
        for r, pr in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    codes[r] = pr.wait()

    def tail(f, nbytes=3000):
        f.flush(); f.seek(0, 2)
        size = f.tell()
        f.seek(max(0, size - nbytes))
        return f.read()
    for r in range(n):                                                          # every rank's stderr reaches the caller
        t = tail(errs[r])
        if t.strip():
            sys.stderr.write("---- bench.py rank %d stderr (tail) ----\n%s\n" % (r, t.rstrip()))
    out0 = tail(outs[0], 1 << 20)
    if out0 and failed is None and not timed_out:
        sys.stdout.write(out0)
        sys.stdout.flush()
    for f in errs + outs:
        f.close()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %s first; ranks failed (rank, exit code): %s; per-rank logs in %s\n"
                         % (failed, codes[failed], [(r, c) for r, c in enumerate(codes) if c != 0], logdir))
        return 1
    if timed_out:
        sys.stderr.write("bench.py: ranks still running after %.0f s were terminated (rank, exit code): %s; per-rank logs in %s\n"
                         % (limit, list(enumerate(codes)), logdir))
        return 1
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    sys.exit(self_launch(_requested_gpus(sys.argv[1:])))

import torch  # noqa: E402

SURVEY_WEIGHT_BYTES = {16: 722.66e6, 32: 790.28e6}      # effective params x 2 B
SURVEY_FLOPS_PER_FACE_STEP = {16: 2.0765e9, 32: 8.2917e9}
HBM_PEAK_GBS = 8000.0                                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0
# rocprofv3 --pmc passes of THIS bench command (tools/collect_profiles.sh + tools/pmc_traffic.py).  The file records the
# hash of the kernel sources it was measured on; `roofline.traffic` is null when that differs from the sources in the tree.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_latest.json")
# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass of this command (tools/collect_mfma.sh + tools/pmc_mfma.py), per latent; same hash gate
MFMA_FILE = os.path.join(ROOT, "profiles", "mfma_latest.json")


# Latency-aware bound of one diffusion step (models/denoiser/model.py:234-261 is a chain of 32 blocks x 5
# dependent GEMM phases + 15 transitions): every term is a measured price with the file it comes from.
LATENCY_TERMS = {
    "boundary_us": 2.21,          # one dependent launch with an empty body: profiles/r04_mid_splitk_mix_bench.txt ("empty launch")
    "handoff_us": 0.8,            # one in-launch hand-off behind an XCD-local flag line: profiles/r03_xcd_barrier_bench.txt
    "stream_TBs": 6.5,            # chip-wide once-read weight stream: profiles/r02_ingest_bench.txt (5.2 default .. 7.1 nt TB/s of unique bytes)
}


def inlaunch_handoffs(latent, launches):
    """Dependent phase-to-phase hand-offs INSIDE the persistent launches of one step (latent 16, batch <= 64): levels 2 / 3 run 4 + 8 + 2 + 2
    blocks x 5 phases in 4 launches (80 phases, 76 hand-offs), levels 0 / 1 run 8 blocks in 4 launches with a halo and a pool exchange each (16)."""
    return 92 if (latent == 16 and launches < 151) else 0


def latency_floor(latent, launches, weight_bytes, flops_launch):
    t = LATENCY_TERMS
    n_h = inlaunch_handoffs(latent, launches)
    b_us = launches * t["boundary_us"]
    h_us = n_h * t["handoff_us"]
    w_us = max(weight_bytes / (t["stream_TBs"] * 1e12), flops_launch / (MFMA_BF16_PEAK_TFLOPS * 1e12)) * 1e6
    return {"latency_floor_ms": round((b_us + h_us + w_us) * 1e-3, 4),
            "latency_floor_terms_us": {"dependent_launches": launches, "launch_boundaries": round(b_us, 1), "inlaunch_handoffs": n_h,
                                       "handoffs": round(h_us, 1), "weight_stream_or_mfma": round(w_us, 1)},
            "latency_floor_sources": {"boundary_us": [t["boundary_us"], "profiles/r04_mid_splitk_mix_bench.txt"],
                                      "handoff_us": [t["handoff_us"], "profiles/r03_xcd_barrier_bench.txt"],
                                      "stream_TBs": [t["stream_TBs"], "profiles/r02_ingest_bench.txt"]}}


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hifidiff_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(P, latent, n_diff, full_config1=True):
    """The CPU oracle (a port of the reference's algorithm, fp32, torch-CPU) timed on this host's cores, bounded:
    (1) BASELINE configs[0] in full when the latent is 16: 1 face, 50-step DDIM, as written (FPG + IDC + denoiser every step,
        models/refiner.py:32-38 inside test_refiner.py:87-91);
    (2) the bench workload's diffusion step at batch 16, as written, >= 10 timed evaluations, extrapolated to n_diff steps;
    (3) the same with the conditioning hoisted out of the loop (denoiser + scheduler only), >= 10 evaluations."""
    from hifidiff_amd import synth
    from oracle import hifidiff_oracle as O
    out = {}
    all_threads = torch.get_num_threads()

    def pick_threads(fn):
        """Small batches do not scale to every core of a large host: take the fastest of {all, 32, 16} threads (one call each)."""
        best = (None, all_threads)
        for n in sorted({all_threads, min(32, all_threads), min(16, all_threads)}, reverse=True):
            torch.set_num_threads(n)
            fn()
            t0 = time.time(); fn(); dt = time.time() - t0
            if best[0] is None or dt < best[0]:
                best = (dt, n)
        torch.set_num_threads(best[1])
        return best[1]

    if full_config1 and latent == 16:
        x1, crl1, crf1 = synth.sample_inputs(1, latent)
        sch = O.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
        sch.set_timesteps(50)
        n1 = pick_threads(lambda: O.refiner_forward(P, x1, torch.full((1,), 500), crf1, crl1))
        t0 = time.time()
        r = O.sample(P, x1, crf1, crl1, sch, "ddim", as_written=True)
        dt = time.time() - t0
        torch.set_num_threads(all_threads)
        out["config1"] = {"value": round(1.0 / dt, 5), "unit": "faces/s", "seconds": round(dt, 3), "cores": n1,
                          "what": "BASELINE configs[0] run in full: 1 face, latent 16, 50-step DDIM eta 0 clip 3.0, as written "
                                  "(FPG+IDC+denoiser every step), final |x| mean %.4f" % float(r.abs().mean())}
    B = 16
    x, crl, crf = synth.sample_inputs(B, latent)
    t = torch.full((B,), 500)

    def timed(fn, min_evals=10, budget=6.0, max_evals=40):
        fn()                                                                   # warm-up
        n, t0 = 0, time.time()
        while n < min_evals or (time.time() - t0 < budget and n < max_evals):
            fn()
            n += 1
        return (time.time() - t0) / n, n

    cores = pick_threads(lambda: O.refiner_forward(P, x, t, crf, crl))
    dt_w, n_w = timed(lambda: O.refiner_forward(P, x, t, crf, crl))
    cond = O.Conditioning(P, crl, crf)
    dt_h, n_h = timed(lambda: O.fused_denoiser(P, x, t, cond=cond))
    torch.set_num_threads(all_threads)
    out["cores"] = cores
    out["as_written"] = {"value": round(B / (dt_w * n_diff), 5), "unit": "faces/s", "s_per_diffusion_step": round(dt_w, 4), "batch": B, "evaluations": n_w}
    out["hoisted"] = {"value": round(B / (dt_h * n_diff), 5), "unit": "faces/s", "s_per_diffusion_step": round(dt_h, 4), "batch": B, "evaluations": n_h}
    return out


def measure_config3(dev, passes=3, reuse=None):
    """BASELINE configs[3] (test_refiner.py:67,162: latent_res = image_res // 8 = 32): batch 64, 32->256 px, 250-step DDIM, one GPU.
    One warm-up pass (captures the graphs, builds the FiLM table) and `passes` timed passes between synchronisations, prologue included,
    like the headline; run AFTER the headline's timed region, reported as `secondary` in the same JSON line."""
    from hifidiff_amd import _lib, sampling, schedulers, synth
    from hifidiff_amd.refiner import FacialRefiner
    import numpy as np
    L, B, n_diff = 32, 64, 250
    P32 = synth.refiner_state_dict(L, reuse=reuse)           # the weights that do not depend on the latent side are the headline's own tensors
    m = FacialRefiner(L)
    m.load_state_dict(P32)
    m.to(dev)
    m.cache_conditioning = False
    x = torch.from_numpy(np.stack([synth.randn(f"x_T/{f}", (4, L, L)) for f in range(B)])).to(dev)
    crl = torch.from_numpy(np.stack([np.float32(0.8) * synth.randn(f"cr_latent/{f}", (4, L, L)) for f in range(B)])).to(dev)
    crf = torch.from_numpy(np.stack([synth.rand(f"cr_face/{f}", (3, 128, 128)) for f in range(B)])).to(dev)
    sch = schedulers.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
    sch.set_timesteps(n_diff)
    Lh = _lib.lib()
    Lh.hd_set_profiling(m.engine.ctx, 1)
    sampling.sample(m, x, crf, crl, sch, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(passes):
        out = sampling.sample(m, x, crf, crl, sch, check=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.check(synchronize=False)
    loop_ms, step_ms_avg = ctypes.c_double(), ctypes.c_double()
    wbytes, fl = ctypes.c_int64(), ctypes.c_double()
    Lh.hd_get_profile(m.engine.ctx, ctypes.byref(loop_ms), ctypes.byref(step_ms_avg), ctypes.byref(wbytes), ctypes.byref(fl))
    step_s = step_ms_avg.value * 1e-3
    flops_launch = SURVEY_FLOPS_PER_FACE_STEP[L] * B
    tflops = flops_launch / step_s / 1e12 if step_s > 0 else 0.0
    gbs = SURVEY_WEIGHT_BYTES[L] / step_s / 1e9 if step_s > 0 else 0.0
    n_launch = Lh.hd_num_ops(m.engine.ctx, 0) * Lh.hd_num_chains(m.engine.ctx)
    roof = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4),
            "algorithmic_flops_per_launch": flops_launch, "algorithmic_bytes_per_launch": SURVEY_WEIGHT_BYTES[L], "avg_launch_ms": round(step_ms_avg.value, 4),
            "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "mfma_busy_frac": None}
    tfile = TRAFFIC_FILE.replace(".json", "_L%d.json" % L)
    if os.path.exists(tfile):
        tj = json.load(open(tfile))
        if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("latent") == L and tj.get("kind") == "ddim":
            roof["traffic"], roof["traffic_source"] = tj["hbm_bytes_per_step"], tj["source"]
    if os.path.exists(MFMA_FILE):
        mj = json.load(open(MFMA_FILE)).get("L%d" % L)
        if mj and mj.get("kernel_source_hash") == kernel_source_hash() and mj.get("kind") == "ddim":
            roof["mfma_busy_frac"], roof["mfma_busy_source"] = round(mj["mfma_busy_frac"], 4), mj["source"]
    roof.update(latency_floor(L, n_launch, SURVEY_WEIGHT_BYTES[L], flops_launch))
    roof["frac_of_latency_floor"] = round(roof["latency_floor_ms"] / step_ms_avg.value, 4) if step_ms_avg.value > 0 else None
    res = {"workload": "BASELINE configs[3]: batch 64, latent 32 (32->256 px), 250-step DDIM (clip 3.0, eta 0), conditioning prologue included, 1 GPU",
           "value": round(B * passes / dt, 3), "unit": "faces/s", "steps": passes, "warmup": 1, "ms_per_step": round(dt / passes * 1e3, 3),
           "ms_per_diffusion_step": round(step_ms_avg.value, 4), "launches_per_diffusion_step": n_launch, "dtype": "bf16", "data": "synthetic",
           "output_finite": bool(torch.isfinite(out).all().item()), "roofline": roof}
    del m
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="faces per GPU")
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--kind", default="ddpm", choices=["ddpm", "ddim"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] passes that follow the headline's timed region")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HD_BENCH_TEST_FAIL_RANK") == str(rank):                 # tests: this rank dies before the rendezvous
        sys.stderr.write("bench.py rank %d/%d: exiting early (HD_BENCH_TEST_FAIL_RANK)\n" % (rank, world))
        raise SystemExit(7)
    if world != a.gpus:
        raise SystemExit("bench.py rank %d/%d: --gpus %d does not match WORLD_SIZE %d" % (rank, world, a.gpus, world))
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py rank %d/%d (local %d, master %s:%s): needs an MI355X (no CPU fallback exists for the product path)\n"
                         % (rank, world, local, os.environ.get("MASTER_ADDR", "-"), os.environ.get("MASTER_PORT", "-")))
        raise SystemExit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("HD_BENCH_FORCE_DIST"))     # FORCE: rehearse the N>1 code path with one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("HD_BENCH_PG_TIMEOUT", "300"))))

    from hifidiff_amd import _lib, distributed, sampling, schedulers, synth
    from hifidiff_amd.refiner import FacialRefiner

    P = synth.refiner_state_dict(a.latent)
    model = FacialRefiner(a.latent)
    model.load_state_dict(P)
    model.to(dev)
    B = a.batch
    # this rank's contiguous slice of the global batch (hifidiff_amd.distributed.shard_range)
    lo, hi = distributed.shard_range(world * B, rank, world)
    x, crl, crf = [], [], []
    import numpy as np
    L = a.latent
    for f in range(lo, hi):
        x.append(synth.randn(f"x_T/{f}", (4, L, L)))
        crl.append(np.float32(0.8) * synth.randn(f"cr_latent/{f}", (4, L, L)))
        crf.append(synth.rand(f"cr_face/{f}", (3, 128, 128)))
    x = torch.from_numpy(np.stack(x)).to(dev)
    crl = torch.from_numpy(np.stack(crl)).to(dev)
    crf = torch.from_numpy(np.stack(crf)).to(dev)

    if a.kind == "ddpm":
        sch = schedulers.DDPMScheduler(clip_sample=True, clip_sample_range=3.0)
        if a.diffusion_steps != 1000:
            sch.set_timesteps(a.diffusion_steps)
    else:
        sch = schedulers.DDIMScheduler(clip_sample=True, clip_sample_range=3.0)
        sch.set_timesteps(a.diffusion_steps)
    n_diff = int(sch.timesteps.numel())
    model.cache_conditioning = False                       # the prologue is part of every timed pass
    Lh = _lib.lib()
    Lh.hd_set_profiling(model.engine.ctx, 1)

    if os.environ.get("HD_DUMP_OPS") and rank == 0:         # op order of one captured step, for tools/prof_summary.py
        model.prepare(crf, crl)
        with open(os.environ["HD_DUMP_OPS"], "w") as f:
            for i in range(Lh.hd_num_ops(model.engine.ctx, 0)):
                f.write(Lh.hd_debug_op_name(model.engine.ctx, 0, i).decode() + "\n")

    def one_pass(seed):
        out = sampling.sample(model, x, crf, crl, sch, noise=None, seed=seed, check=False)   # the timed region has its own fence; model.check() after it
        return distributed.gather_faces(out, world * B)    # result gather over RCCL/xGMI (4 KB per face); identity at N=1

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        one_pass(1000 + w)
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_pass(k + rank * 7919)
    fence()
    dt = time.perf_counter() - t0
    model.check(synchronize=False)                         # a persistent stage that gave up (results NaN) fails the run here
    # HIP-event time of the last pass's replay loop
    loop_ms, step_ms_avg = ctypes.c_double(), ctypes.c_double()
    wbytes, fl = ctypes.c_int64(), ctypes.c_double()
    Lh.hd_get_profile(model.engine.ctx, ctypes.byref(loop_ms), ctypes.byref(step_ms_avg), ctypes.byref(wbytes), ctypes.byref(fl))
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = torch.tensor([hi - lo], dtype=torch.int64, device=dev)
    rccl_world = 1
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        counts = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(counts, per_rank)
        per_rank = torch.cat(counts)
        rccl_world = dist.get_world_size()
    dt = float(tt.item())
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        faces_per_rank = [int(v) for v in per_rank.tolist()]
        faces = sum(faces_per_rank) * a.steps
        value = faces / dt
        alg_bytes = SURVEY_WEIGHT_BYTES.get(a.latent, float(wbytes.value))
        step_s = step_ms_avg.value * 1e-3
        achieved = alg_bytes / step_s / 1e9 if step_s > 0 else 0.0
        flops_launch = SURVEY_FLOPS_PER_FACE_STEP.get(L, fl.value) * B
        tflops = flops_launch / step_s / 1e12 if step_s > 0 else 0.0
        traffic, traffic_src, measured_gbs = None, None, None
        tfile = TRAFFIC_FILE if a.latent == 16 else TRAFFIC_FILE.replace(".json", "_L%d.json" % a.latent)   # one file per latent
        if os.path.exists(tfile) and B == 64 and world == 1:
            tj = json.load(open(tfile))
            if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("latent") == a.latent and tj.get("kind") == a.kind:
                traffic, traffic_src = tj["hbm_bytes_per_step"], tj["source"]
                measured_gbs = round(traffic / step_s / 1e9, 1) if step_s > 0 else None
        mfma_busy, mfma_src = None, None
        if os.path.exists(MFMA_FILE) and B == 64 and world == 1:
            mj = json.load(open(MFMA_FILE)).get("L%d" % a.latent)
            if mj and mj.get("kernel_source_hash") == kernel_source_hash() and mj.get("kind") == a.kind:
                mfma_busy, mfma_src = round(mj["mfma_busy_frac"], 4), mj["source"]
        headline = (L == 16 and a.kind == "ddpm" and n_diff == 1000 and B == 64)
        cfg3 = (L == 32 and a.kind == "ddim" and n_diff == 250 and B == 64)
        # which roofline bounds a step: weight streaming (HBM) at latent 16, MFMA at latent 32
        hbm_bound = (alg_bytes / (HBM_PEAK_GBS * 1e9)) >= (flops_launch / (MFMA_BF16_PEAK_TFLOPS * 1e12))
        if hbm_bound:
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)}
        else:
            roof = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4)}
        roof.update({"traffic": traffic, "traffic_source": traffic_src, "achieved_from_measured_traffic_GBs": measured_gbs,
                     "kernel": "one captured step graph = one denoiser evaluation of the batch, scheduler update fused into its last launch",
                     "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops_launch,
                     "packed_weight_bytes_counted_by_library": int(wbytes.value),
                     "avg_launch_ms": round(step_ms_avg.value, 4),
                     "hbm_frac": round(achieved / HBM_PEAK_GBS, 4), "mfma_frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                     "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_src})
        n_launch = Lh.hd_num_ops(model.engine.ctx, 0) * Lh.hd_num_chains(model.engine.ctx)
        roof.update(latency_floor(a.latent, n_launch, alg_bytes, flops_launch))
        roof["frac_of_latency_floor"] = round(roof["latency_floor_ms"] / step_ms_avg.value, 4) if step_ms_avg.value > 0 else None
        res = {
            "metric": "faces/sec (whole node), 16→128 1000-step reverse diffusion, batch 64",
            "value": round(value, 3), "unit": "faces/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%sbatch %d per GPU, latent %d (%d->%d px), %d-step %s "
                                   "(clip 3.0%s), conditioning prologue (FPG, ResNet-50 IDC, HCA gates, idc_conv) included%s"
                                   % ("BASELINE configs[1]: " if (headline and world == 1) else
                                      "BASELINE configs[2]: " if (headline and world == 8) else
                                      "BASELINE configs[4] (IDC forward timed): " if (headline and world == 4) else
                                      "BASELINE configs[3]: " if cfg3 else "",
                                      B, L, L, L * 8, n_diff, a.kind.upper(), ", fixed_small variance" if a.kind == "ddpm" else ", eta 0",
                                      ", device Philox noise" if a.kind == "ddpm" else ""),
                       "faces_per_gpu": B, "faces_per_rank": faces_per_rank, "rccl_world_size": rccl_world,
                       "launched_by": "bench.py self-launch" if os.environ.get("HD_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "single process"),
                       "latent_res": L, "diffusion_steps": n_diff, "sampler": a.kind,
                       "parallelism": "batch-sharded x%d, no in-loop collective, one all_gather of the latents per pass" % world,
                       "concurrent_chains": Lh.hd_num_chains(model.engine.ctx),
                       "launches_per_diffusion_step": Lh.hd_num_ops(model.engine.ctx, 0) * Lh.hd_num_chains(model.engine.ctx),
                       "output_finite": finite},
            "roofline": roof,
        }
        if headline and world == 1 and not a.no_secondary:    # BASELINE configs[3], after the headline's timed region; never touches `value`
            res["secondary"] = measure_config3(dev, reuse=(P, a.latent))
        if not a.no_cpu_baseline and world == 1:              # the CPU leg is reported at N=1 only
            cb = cpu_baseline(P, a.latent, n_diff)
            res["cpu_baseline"] = {
                "value": cb["as_written"]["value"], "unit": "faces/s", "cores": cb["cores"], "host_cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle (torch-CPU fp32 port of the reference), the bench workload's diffusion step as written "
                          "(FPG+IDC+denoiser per step, models/refiner.py:32-38), batch %d, %d timed evaluations after 1 warm-up: "
                          "%.3f s per step, extrapolated x%d steps (per-step cost is constant)"
                          % (cb["as_written"]["batch"], cb["as_written"]["evaluations"], cb["as_written"]["s_per_diffusion_step"], n_diff),
                "hoisted": cb["hoisted"], "config1_full": cb.get("config1")}
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
