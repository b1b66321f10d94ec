#!/usr/bin/env python3
"""Headline benchmark: faces/sec, 16->128 px, 1000-step DDPM reverse diffusion, batch 64 per GPU
(BASELINE.json configs[1]; SURVEY §8d "Config 2").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete reverse-diffusion pass over the rank's batch of synthetic faces: the
once-per-batch conditioning prologue (FPG, ResNet-50 IDC, HCA gates, idc_conv), the FiLM table for all
timesteps, and `diffusion_steps` graph-replayed denoiser evaluations + scheduler updates, inputs already
resident in HBM.  Faces are independent, so ranks shard the batch with no collective inside the loop;
RCCL is used only for the final gather of the latents (inside the timed region) and the timing reduce.

Prints ONE JSON line (rank 0).  `roofline`: one launch = one captured step graph (one denoiser
evaluation of the whole batch); algorithmic bytes per launch = the bf16 weights every step must stream
(0.7227 GB at latent 16), duration = HIP-event time of the replay loop / diffusion steps.
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
import torch  # noqa: E402

SURVEY_WEIGHT_BYTES = {16: 722.66e6, 32: 790.28e6}      # effective params x 2 B
SURVEY_FLOPS_PER_FACE_STEP = {16: 2.0765e9, 32: 8.2917e9}
HBM_PEAK_GBS = 8000.0                                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_traffic.json")   # rocprofv3 --pmc passes (tools/pmc_traffic.py)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def cpu_baseline(P, latent, kind, threads_note=True):
    """Oracle timed on the host: batch 16, as-written forward (FPG + IDC + denoiser per step), fp32."""
    from hifidiff_amd import synth
    from oracle import hifidiff_oracle as O
    B = 16
    x, crl, crf = synth.sample_inputs(B, latent)
    t = torch.full((B,), 500)
    O.refiner_forward(P, x, t, crf, crl)                  # warm-up
    n, t0 = 0, time.time()
    while True:
        eps = O.refiner_forward(P, x, t, crf, crl)
        n += 1
        if time.time() - t0 > 10.0 or n >= 20:
            break
    dt = (time.time() - t0) / n
    return dt, B, n, float(eps.abs().mean())


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
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("HD_BENCH_FORCE_DIST"))     # FORCE: rehearse the N>1 code path with one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hifidiff_amd import _lib, sampling, schedulers, synth
    from hifidiff_amd.refiner import FacialRefiner

    P = synth.refiner_state_dict(a.latent)
    model = FacialRefiner(a.latent)
    model.load_state_dict(P)
    model.to(dev)
    B = a.batch
    # this rank's faces of the global batch
    x, crl, crf = [], [], []
    import numpy as np
    L = a.latent
    for f in range(rank * B, rank * B + B):
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

    gathered = [torch.empty_like(x) for _ in range(world)] if use_dist else None
    if os.environ.get("HD_DUMP_OPS") and rank == 0:         # op order of one captured step, for tools/prof_summary.py
        model.prepare(crf, crl)
        with open(os.environ["HD_DUMP_OPS"], "w") as f:
            for i in range(Lh.hd_num_ops(model.engine.ctx, 0)):
                f.write(Lh.hd_debug_op_name(model.engine.ctx, 0, i).decode() + "\n")

    def one_pass(seed):
        out = sampling.sample(model, x, crf, crl, sch, noise=None, seed=seed)
        if use_dist:
            dist.all_gather(gathered, out)                 # result gather over RCCL/xGMI (4 KB per face)
        return out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        one_pass(1000 + w)
    fence()
    t0 = time.perf_counter()
    step_ms = []
    for k in range(a.steps):
        out = one_pass(k + rank * 7919)
    fence()
    dt = time.perf_counter() - t0
    # HIP-event time of the last pass's replay loop
    loop_ms, step_ms_avg = ctypes.c_double(), ctypes.c_double()
    wbytes, fl = ctypes.c_int64(), ctypes.c_double()
    Lh.hd_get_profile(model.engine.ctx, ctypes.byref(loop_ms), ctypes.byref(step_ms_avg), ctypes.byref(wbytes), ctypes.byref(fl))
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        faces = world * B * a.steps
        value = faces / dt
        alg_bytes = SURVEY_WEIGHT_BYTES.get(a.latent, float(wbytes.value))
        step_s = step_ms_avg.value * 1e-3
        achieved = alg_bytes / step_s / 1e9 if step_s > 0 else 0.0
        traffic, traffic_src = None, None
        if os.path.exists(TRAFFIC_FILE) and a.latent == 16 and B == 64 and a.kind == "ddpm":
            tj = json.load(open(TRAFFIC_FILE))
            traffic, traffic_src = tj["hbm_bytes_per_step"], tj["source"]
        res = {
            "metric": "faces/sec (whole node), 16→128 1000-step reverse diffusion, batch 64",
            "value": round(value, 3), "unit": "faces/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%sbatch %d per GPU, latent %d (%d->%d px), %d-step %s "
                                   "(clip 3.0%s), conditioning prologue included%s"
                                   % ("BASELINE configs[1]: " if (L == 16 and a.kind == "ddpm" and n_diff == 1000 and B == 64) else
                                      "BASELINE configs[3]: " if (L == 32 and a.kind == "ddim" and n_diff == 250 and B == 64) else "",
                                      B, L, L, L * 8, n_diff, a.kind.upper(), ", fixed_small variance" if a.kind == "ddpm" else ", eta 0",
                                      ", device Philox noise" if a.kind == "ddpm" else ""),
                       "faces_per_gpu": B, "latent_res": L, "diffusion_steps": n_diff, "sampler": a.kind,
                       "parallelism": "batch-sharded x%d, no in-loop collective" % world,
                       "concurrent_chains": Lh.hd_num_chains(model.engine.ctx),
                       "launches_per_diffusion_step": Lh.hd_num_ops(model.engine.ctx, 0) * Lh.hd_num_chains(model.engine.ctx),
                       "output_finite": finite},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "one captured step graph = one denoiser evaluation of the batch, scheduler update fused into its last launch",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "packed_weight_bytes_counted_by_library": int(wbytes.value),
                         "avg_launch_ms": round(step_ms_avg.value, 4),
                         "mfma_frac": round(SURVEY_FLOPS_PER_FACE_STEP.get(L, fl.value) * B / step_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if step_s > 0 else None},
        }
        if not a.no_cpu_baseline and world == 1:              # the CPU leg is reported at N=1 only
            sdt, sb, sn, _ = cpu_baseline(P, a.latent, a.kind)
            res["cpu_baseline"] = {
                "value": round(sb / (sdt * n_diff), 5), "unit": "faces/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle refiner_forward as written (FPG+IDC+denoiser per step, fp32, torch-CPU), batch %d, "
                          "%d evaluations after 1 warm-up: %.3f s per diffusion step, extrapolated x%d steps"
                          % (sb, sn, sdt, n_diff)}
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
