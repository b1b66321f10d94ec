"""Phase-by-phase comparison of an XCD-local persistent stage (hd_xcd.hpp) with the per-GEMM launches of the same blocks:
model A is built with HD_NO_XCD=1 (151 launches, one per GEMM), model B with the stages; for every phase n of every stage B
stops after n phases, A after the matching launch, and the hand-off buffers are compared bit for bit."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifidiff_amd import _lib, synth  # noqa: E402
from hifidiff_amd.refiner import FacialRefiner  # noqa: E402

torch.set_grad_enabled(False)
L = _lib.lib()
L.hd_debug_read.restype = ctypes.c_int64


def read(m, name):
    n = L.hd_debug_read(m.engine.ctx, name.encode(), None, 0)
    _lib.check(n, m.engine.ctx)
    buf = np.empty(n, dtype=np.float32)
    _lib.check(L.hd_debug_read(m.engine.ctx, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), n), m.engine.ctx)
    return buf


def names(m):
    return [L.hd_debug_op_name(m.engine.ctx, 0, i).decode() for i in range(L.hd_num_ops(m.engine.ctx, 0))]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    W = synth.refiner_state_dict(16)
    os.environ["HD_NO_XCD"] = "1"
    a = FacialRefiner(16); a.load_state_dict(W); a.to("cuda:0")
    del os.environ["HD_NO_XCD"]
    b = FacialRefiner(16); b.load_state_dict(W); b.to("cuda:0")
    x, crl, crf = [t.cuda() for t in synth.sample_inputs(B, 16)]
    a(x, 500, crf, crl); b(x, 500, crf, crl)
    na, nb = names(a), names(b)
    print(len(na), "launches vs", len(nb))
    stages = [("denoiser.encoders.2", 4, 2), ("denoiser.encoders.3", 8, 3), ("denoiser.decoders.0", 2, 3), ("denoiser.decoders.1", 2, 2)]
    suffix = ["conv2_gate_pool", "sca", "conv3", "conv4", "conv5"]
    watch = {0: ["pooled16_"], 1: ["G", "S"], 2: ["Yb", "sy"], 3: ["G"], 4: ["Xb", "sx"]}
    first_bad = None
    for pref, nblk, lvl in stages:
        if only and only not in pref:
            continue
        stage_op = nb.index(f"{pref}.{nblk - 1}.conv5")
        for n in range(1, 5 * nblk + 1):
            blk, q = (n - 1) // 5, (n - 1) % 5
            ia = na.index(f"{pref}.{blk}.{suffix[q]}")
            _lib.check(L.hd_debug_limit_ops(a.engine.ctx, 0, ia + 1)); _lib.check(L.hd_debug_limit_ops(b.engine.ctx, 0, stage_op + 1))
            _lib.check(L.hd_set_option(b.engine.ctx, b"xcd_phase_limit", n))
            a(x, 500, crf, crl); b(x, 500, crf, crl)
            last_gated = (n == 5 * nblk and "decoders" in pref)
            bufs = (["Xg", "X"] if last_gated else watch[q]) + (["X"] if (q == 4 and not last_gated and n == 5 * nblk) else [])
            for k in bufs:
                va, vb = read(a, f"{k}{lvl}"), read(b, f"{k}{lvl}")
                neq = int((va.view(np.uint32) != vb.view(np.uint32)).sum())
                if neq or "-v" in sys.argv:
                    idx = np.nonzero(va.view(np.uint32) != vb.view(np.uint32))[0]
                    print(f"{pref}.{blk}.{suffix[q]:16s} {k}{lvl}: {neq}/{va.size} words differ, max abs {np.abs(va - vb).max():.3e}, first at {idx[:6].tolist()}", flush=True)
                    if first_bad is None and neq:
                        first_bad = (pref, blk, q, k)
        _lib.check(L.hd_set_option(b.engine.ctx, b"xcd_phase_limit", 0))
    print("first difference:", first_bad)


if __name__ == "__main__":
    main()
