#!/usr/bin/env python3
"""Stage-by-stage parity of the HIP CoarseRestoration against the CPU oracle (bf16-emulation mode) on the GPU box.
usage: python tools/cr_parity.py [--batch 2]   (test infrastructure: uses oracle/)"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifidiff_amd import _lib, synth                      # noqa: E402
from hifidiff_amd.cr import CoarseRestoration             # noqa: E402
from oracle import hifidiff_oracle as O                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    P = synth.cr_state_dict()
    x = torch.from_numpy(np.stack([synth.rand(f"ln_face/{f}", (3, 128, 128)) for f in range(a.batch)]))
    taps = {}
    ref = O.coarse_restoration(P, x, prec=O.BF16, taps=taps)
    m = CoarseRestoration()
    m.load_state_dict(P)
    m.to("cuda:0")
    xd = x.cuda()
    out = m(xd).cpu()
    L = _lib.lib()
    n = L.hd_num_ops(m._ctx, 0)
    for i in range(n):
        name = L.hd_debug_op_name(m._ctx, 0, i).decode()
        if name not in taps:
            continue
        L.hd_debug_limit_ops(m._ctx, 0, i + 1)
        m(xd)
        cnt = L.hd_debug_read_op(m._ctx, 0, i, None, 0)
        buf = np.empty(cnt, dtype=np.float32)
        _lib.check(L.hd_debug_read_op(m._ctx, 0, i, buf.ctypes.data_as(ctypes.c_void_p), cnt), m._ctx)
        t = taps[name]
        r = t.reshape(-1) if (name in ("outro",) or t.dim() != 4) else t.permute(0, 2, 3, 1).reshape(-1)
        got = torch.from_numpy(buf)
        if got.numel() != r.numel():
            print(f"{i:4d} {name:40s} size {got.numel()} vs {r.numel()}")
            continue
        rel = float((got.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))
        print(f"{i:4d} {name:40s} rel {rel:.3e}")
    L.hd_debug_limit_ops(m._ctx, 0, -1)
    print("final rel vs oracle(bf16):", float((out.double() - ref.double()).norm() / ref.double().norm()))


if __name__ == "__main__":
    main()
