"""Find the first launch of the denoiser program whose output differs between two identical runs (GPU).

usage: python tools/determinism_scan.py [batch] [latent] [repeats] [which: 0 denoiser, 1 conditioning prologue] [per-face timesteps: 1 / 0]
Every launch is meant to be bitwise reproducible (fixed reduction orders, no float atomics); a launch that is not has a
race or reads something uninitialised.  Runs the program up to launch i twice (hd_debug_limit_ops) and compares its output.
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hifidiff_amd import _lib, refiner, synth
from tools.op_parity import read_op


def scan(B=64, latent=16, reps=3, model=None, verbose=True, max_bad=5, which=0, per_face=True, only=None):
    """Returns (launches scanned, [(index, name, differing values, max abs difference), ...]).
    which = 0: the denoiser program (one eps evaluation); 1: the conditioning prologue (FPG, ResNet-50 IDC, gates).
    only: predicate on the launch name (None: every launch); the return count is then the number of launches it selected.
    Repeats beyond the second are compared with the first run and dropped (300 repeats keep two copies only).
    per_face: a timestep per face (the LayerNorm GEMMs read FiLM rows of the global table: LdF32LNFace) or one for all faces
    (the sampling loop's form: the shared row copied to LDS, LdF32LN) -- different kernels."""
    m = model
    if m is None:
        m = refiner.FacialRefiner(latent)
        m.load_state_dict(synth.refiner_state_dict(latent))
        m.to("cuda")
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, latent)]
    e = m.engine
    e.prepare(crl, cr_face=crf)
    L = _lib.lib()
    t = torch.full((B,), 500.0, device="cuda") if per_face else 500.0
    n = L.hd_num_ops(e.ctx, which)
    bad = []
    scanned = 0
    for i in range(n):
        name = L.hd_debug_op_name(e.ctx, which, i).decode()
        if only is not None and not only(name):
            continue
        scanned += 1
        L.hd_debug_limit_ops(e.ctx, which, i + 1)
        outs = []
        for r in range(reps):
            if which == 1:
                e.prepare(crl, cr_face=crf)
            else:
                e.eps(x, t)
            torch.cuda.synchronize()
            o = read_op(L, e.ctx, which, i).numpy()
            if r < 2 or not np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)):
                outs.append(o.copy())
                if r >= 2:
                    break
        same = all(np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs[1:])
        if not same:
            d = max(float(np.abs(outs[0] - o).max()) for o in outs[1:])
            nd = max(int((outs[0].view(np.uint32) != o.view(np.uint32)).sum()) for o in outs[1:])
            if verbose:
                print(f"{i:3d} {name:45s} DIFFERS: {nd} of {outs[0].size} values, max abs {d:.3e}", flush=True)
            bad.append((i, name, nd, d))
            if len(bad) >= max_bad:
                break
    L.hd_debug_limit_ops(e.ctx, which, -1)
    if which == 1:
        e.prepare(crl, cr_face=crf)
    return (scanned if only is not None else n), bad


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    latent = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    which = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    per_face = (int(sys.argv[5]) != 0) if len(sys.argv) > 5 else True
    n, bad = scan(B, latent, reps, which=which, per_face=per_face)
    print(f"{n} launches scanned, {len(bad)} not reproducible")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
