#!/usr/bin/env python3
"""Teacher-forced op parity: every launch of one denoiser evaluation against the CPU oracle ON THE LAUNCH'S OWN INPUTS.

The chained scan (tools/op_parity.py) compares launch i with the oracle's tap i; both sides then carry the drift of everything
before them (the bf16-operand emulation is chaotic at the level of operand rounding: DESIGN.md §2), so its bound at the deep
levels is 2e-2 and a 1-2 % kernel error would pass.  Here the state the HIP path itself has reached BEFORE launch i is read
back (hd_debug_read), the oracle's arithmetic for that one launch is applied to exactly those values on the CPU (bf16-operand
emulation: same rounding points), and the launch's output is compared with that: what is left is accumulation order and the
few rounding flips it causes.  Bounds: fp32 outputs 3e-4, bf16-stored outputs 3e-3 (rel-L2), at batch 2 and at the benchmark
batch 64 (the tile shapes differ).  Needs the program with one launch per GEMM (model created under HD_NO_XCD=1); the
XCD-local persistent stages are tied to these launches bit for bit by their own test.  (Test infrastructure: uses oracle/.)
"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hifidiff_amd import _lib                                  # noqa: E402
from oracle import hifidiff_oracle as O                         # noqa: E402

PR = O.BF16


def _read(L, ctx, name):
    L.hd_debug_read.restype = ctypes.c_int64
    n = L.hd_debug_read(ctx, name.encode(), None, 0)
    _lib.check(n, ctx)
    buf = np.empty(n, dtype=np.float32)
    _lib.check(L.hd_debug_read(ctx, name.encode(), buf.ctypes.data_as(ctypes.c_void_p), n), ctx)
    return torch.from_numpy(buf)


def _nchw(flat, B, C, H):
    return flat[:B * H * H * C].reshape(B, H, H, C).permute(0, 3, 1, 2).contiguous()


def _rows(t):
    return t.permute(0, 2, 3, 1).reshape(-1) if t.dim() == 4 else t.reshape(-1)


def _rel(got, want):
    d = got.double() - want.double()
    return float(d.norm() / want.double().norm().clamp_min(1e-30)), float(d.abs().max())


def level_of(name, latent):
    p = name.split(".")
    if p[1] == "encoders":
        l = int(p[2])
    elif p[1] == "middle_blks":
        l = 4
    else:
        l = 3 - int(p[2])
    return l, 128 << l, latent >> l


def forced_scan(model, P, x, crl, crf, t, report, fp32_bound=3e-4, bf16_bound=3e-3):
    """model: FacialRefiner built under HD_NO_XCD=1 (151 launches at latent 16).  Returns the worst rel-L2 over fp32 outputs and
    over bf16-stored outputs; report lines carry `<<<<<<` where a bound is exceeded."""
    L = _lib.lib()
    e = model.engine
    ctx = e.ctx
    B, latent = x.shape[0], e.latent_res
    s = latent // 16
    e.prepare(crl.cuda(), cr_face=crf.cuda())
    xd = x.cuda()
    n = L.hd_num_ops(ctx, 0)
    names = [L.hd_debug_op_name(ctx, 0, i).decode() for i in range(n)]
    tt = O.normalize_timesteps(t, B)
    temb = O.time_embedding(P, tt)
    # conditioning the HIP path itself computed (its parity is the prologue scan's business)
    idc = _read(L, ctx, "idc")
    gate_c = {i: _read(L, ctx, f"wc{i}") for i in range(5)}
    gate_s = {i: _read(L, ctx, f"ws{i}") for i in range(5)}
    worst = {"fp32": 0.0, "bf16": 0.0}

    def run_to(i):
        L.hd_debug_limit_ops(ctx, 0, i)
        e.eps(xd, t)

    def check(i, name, what, got, want, stored_bf16):
        rel, mx = _rel(got, want)
        kind = "bf16" if stored_bf16 else "fp32"
        lim = bf16_bound if stored_bf16 else fp32_bound
        worst[kind] = max(worst[kind], rel if rel == rel else 1e9)
        report.append(f"{i:3d} {name:42s} {what:8s} rel {rel:.3e} maxabs {mx:.3e} ({kind} <= {lim:.0e}){'' if rel <= lim else '  <<<<<<'}")

    gated_last = {"denoiser.middle_blks.7": 0, "denoiser.decoders.0.1": 1, "denoiser.decoders.1.1": 2, "denoiser.decoders.2.1": 3, "denoiser.decoders.3.1": 4}
    for i, name in enumerate(names):
        parts = name.split(".")
        kind = parts[-1] if parts[0] == "denoiser" else parts[0]
        # ---- state before the launch ----
        run_to(i)
        if parts[0] == "denoiser":
            p = ".".join(parts[:-1])
            l, C, H = level_of(name, latent)
            sl = str(l)
            film = O.film_vectors(P, p, temb)
            M = B * H * H
            def gate_of_x():
                # conv1 -> depthwise 3x3 -> SimpleGate on the HIP path's own block input (X is not written before conv5)
                inp = _nchw(_read(L, ctx, "X" + sl), B, C, H)
                h = O.layernorm2d(inp, P[p + ".norm1.weight"], P[p + ".norm1.bias"], prec=PR) * (film[1] + 1) + film[0]
                t1 = O._gemm_conv(h, P[p + ".conv1.weight"], P[p + ".conv1.bias"], PR)
                return O.simple_gate(F.conv2d(t1, P[p + ".conv2.weight"], P[p + ".conv2.bias"], padding=1, groups=2 * C))
            if kind == "conv1":
                # unfused form (faces too large for the fused depthwise epilogue: latent 32, level 0): T1 has no tap of its own, the
                # next launch's G is checked against the oracle's conv1 -> depthwise -> gate of the same input
                report.append(f"{i:3d} {name:42s} (checked through the next launch's G)")
            elif kind == "conv2_gate_pool":
                unfused = names[i - 1].endswith(".conv1") or (H, C) in ((32, 128), (16, 256))   # or by strips (hd_strip.hpp): the next launch adds the strip sums up
                g = gate_of_x()
                run_to(i + 1)
                check(i, name, "G", _read(L, ctx, "G" + sl)[:M * C], _rows(PR.q(g)), True)
                if not unfused:                                       # unfused: band sums only, the mean is pool_finish's output
                    check(i, name, "pooled", _read(L, ctx, "pooled" + sl)[:B * C], g.mean(dim=(2, 3)).reshape(-1), False)
            elif kind == "pool_finish":
                g = gate_of_x()
                run_to(i + 1)
                check(i, name, "pooled", _read(L, ctx, "pooled" + sl)[:B * C], g.mean(dim=(2, 3)).reshape(-1), False)
            elif kind == "sca":
                pooled = _read(L, ctx, "pooled" + sl)[:B * C].reshape(B, C, 1, 1)
                g = _nchw(_read(L, ctx, "G" + sl), B, C, H)
                run_to(i + 1)
                sv = O._gemm_conv(pooled, P[p + ".sca.1.weight"], P[p + ".sca.1.bias"], PR)
                check(i, name, "S", _read(L, ctx, "S" + sl)[:B * C], sv.reshape(-1), False)
                if H * H <= 16:                                       # the launch also rescales G in place (few pixels per face)
                    check(i, name, "G*s", _read(L, ctx, "G" + sl)[:M * C], _rows(PR.q(g * sv)), True)
            elif kind == "conv3":
                g = _nchw(_read(L, ctx, "G" + sl), B, C, H)
                inp = _nchw(_read(L, ctx, "X" + sl), B, C, H)
                if H * H > 16:                                        # G is scaled by the loader instead
                    g = PR.q(g * _read(L, ctx, "S" + sl)[:B * C].reshape(B, C, 1, 1))
                run_to(i + 1)
                y = inp + O._gemm_conv(g, P[p + ".conv3.weight"], P[p + ".conv3.bias"], PR) * P[p + ".beta"]
                check(i, name, "Y", _read(L, ctx, "Y" + sl)[:M * C], _rows(y), False)
            elif kind == "conv4":
                y = _nchw(_read(L, ctx, "Y" + sl), B, C, H)
                run_to(i + 1)
                h = O.layernorm2d(y, P[p + ".norm2.weight"], P[p + ".norm2.bias"], prec=PR) * (film[3] + 1) + film[2]
                g2 = O.simple_gate(O._gemm_conv(h, P[p + ".conv4.weight"], P[p + ".conv4.bias"], PR))
                check(i, name, "G2", _read(L, ctx, "G" + sl)[:M * C], _rows(PR.q(g2)), True)
            elif kind == "conv5":
                fused = names[i - 1].endswith((".conv2_gate_pool", ".pool_finish"))      # levels 0/1: sca .. conv5 in one launch (hd_chain.hpp)
                g = _nchw(_read(L, ctx, "G" + sl), B, C, H)
                by_strips = fused and (H, C) in ((32, 128), (16, 256))          # hd_strip.hpp left per-strip sums: this launch adds them up and stores the mean
                if fused:
                    inp = _nchw(_read(L, ctx, "X" + sl), B, C, H)
                    if by_strips:
                        want_pool = gate_of_x().mean(dim=(2, 3)).reshape(-1)
                    else:
                        pooled = _read(L, ctx, "pooled" + sl)[:B * C].reshape(B, C, 1, 1)
                else:
                    y = _nchw(_read(L, ctx, "Y" + sl), B, C, H)
                run_to(i + 1)
                if by_strips:
                    pooled = _read(L, ctx, "pooled" + sl)[:B * C]
                    check(i, name, "pooled", pooled, want_pool, False)
                    pooled = pooled.reshape(B, C, 1, 1)
                if fused:
                    sv = O._gemm_conv(pooled, P[p + ".sca.1.weight"], P[p + ".sca.1.bias"], PR)
                    y = inp + O._gemm_conv(PR.q(g * sv), P[p + ".conv3.weight"], P[p + ".conv3.bias"], PR) * P[p + ".beta"]
                    h = O.layernorm2d(y, P[p + ".norm2.weight"], P[p + ".norm2.bias"], prec=PR) * (film[3] + 1) + film[2]
                    g = PR.q(O.simple_gate(O._gemm_conv(h, P[p + ".conv4.weight"], P[p + ".conv4.bias"], PR)))
                out = y + O._gemm_conv(g, P[p + ".conv5.weight"], P[p + ".conv5.bias"], PR) * P[p + ".gamma"]
                check(i, name, "X", _read(L, ctx, "X" + sl)[:M * C], _rows(out), False)
                if model.engine.conditional and p in gated_last:     # the launch also emits the HCA conv's gated input
                    gi = gated_last[p]
                    add = _nchw(idc, B, C, H) if gi == 0 else 0.0
                    wc = gate_c[gi][:B * C].reshape(B, C, 1, 1)
                    ws = gate_s[gi][:M].reshape(B, 1, H, H)
                    hip_x = _nchw(_read(L, ctx, "X" + sl), B, C, H)    # its own fp32 output: the gate multiply alone is checked here
                    check(i, name, "Xg", _read(L, ctx, "Xg" + sl)[:M * C], _rows(PR.q((hip_x + add) * (1.0 + wc + ws))), True)
            else:
                report.append(f"{i:3d} {name:42s} (no rule)")
        elif kind == "intro":
            run_to(i + 1)
            want = F.conv2d(x, P["denoiser.intro.weight"], P["denoiser.intro.bias"], padding=1)
            check(i, name, "X0", _read(L, ctx, "X0")[:B * latent * latent * 128], _rows(want), False)
        elif kind == "downs":
            l = int(parts[1]); C, H = 128 << l, latent >> l
            inp = _nchw(_read(L, ctx, "X" + str(l)), B, C, H)
            run_to(i + 1)
            want = O._gemm_conv(inp, P[f"denoiser.downs.{l}.weight"], P[f"denoiser.downs.{l}.bias"], PR, stride=2)
            check(i, name, "X", _read(L, ctx, "X" + str(l + 1))[:want.numel()], _rows(want), False)
        elif kind == "ups":
            k = int(parts[1]); hi, lo = 4 - k, 3 - k
            Ch, Hh, Cl, Hl = 128 << hi, latent >> hi, 128 << lo, latent >> lo
            src = _nchw(_read(L, ctx, ("Y" if model.engine.conditional else "X") + str(hi)), B, Ch, Hh)
            skip = _nchw(_read(L, ctx, "X" + str(lo)), B, Cl, Hl)
            run_to(i + 1)
            want = O._up_shuffle(src, P[f"denoiser.ups.{k}.0.weight"], 2, PR) + skip
            check(i, name, "X", _read(L, ctx, "X" + str(lo))[:want.numel()], _rows(want), False)
        elif kind == "hcas":
            k = int(parts[1]); l = 4 - k; C, H = 128 << l, latent >> l
            xg = _nchw(_read(L, ctx, "Xg" + str(l)), B, C, H)
            run_to(i + 1)
            q = f"denoiser.hcas.{k}"
            want = torch.relu(O._conv_bn(xg, P, q + ".fused_mlp.0", q + ".fused_mlp.1", PR, padding=1))
            check(i, name, "Y", _read(L, ctx, "Y" + str(l))[:want.numel()], _rows(want), False)
        elif kind == "ending":
            src = _nchw(_read(L, ctx, ("Y0" if model.engine.conditional else "X0")), B, 128, latent)
            run_to(i + 1)
            want = F.conv2d(src, P["denoiser.ending.weight"], P["denoiser.ending.bias"], padding=1)
            check(i, name, "eps", _read(L, ctx, "eps")[:want.numel()], want.reshape(-1), False)
        else:
            report.append(f"{i:3d} {name:42s} (no rule)")
    L.hd_debug_limit_ops(ctx, 0, -1)
    return worst


def stage_forced_scan(model, P, x, crl, crf, t, report, fp32_bound=3e-4, bf16_bound=3e-3):
    """The persistent stages of the DEFAULT program (59 launches at latent 16, batch <= 64: hd_face.hpp levels 0 / 1, hd_xcd.hpp /
    hd_xcd2.hpp levels 2 / 3), block by block: the stage is stopped after b blocks (`face_block_limit` / `xcd_phase_limit`
    = 5 b), the residual stream it has reached is read back, the oracle's arithmetic for block b + 1 alone
    (conditional_naf.py:108-136, bf16-operand emulation) is applied to exactly those values and compared with what the stage
    holds after b + 1 blocks.  The stages therefore need not share a single bit with the per-GEMM launches.  At the end of a
    stage the bf16 copy / the gated HCA input it emits are checked against the stage's own fp32 output."""
    L = _lib.lib()
    e = model.engine
    ctx = e.ctx
    B, latent = x.shape[0], e.latent_res
    e.prepare(crl.cuda(), cr_face=crf.cuda())
    xd = x.cuda()
    n = L.hd_num_ops(ctx, 0)
    names = [L.hd_debug_op_name(ctx, 0, i).decode() for i in range(n)]
    tt = O.normalize_timesteps(t, B)
    temb = O.time_embedding(P, tt)
    idc = _read(L, ctx, "idc")
    gate_c = {i: _read(L, ctx, f"wc{i}") for i in range(5)}
    gate_s = {i: _read(L, ctx, f"ws{i}") for i in range(5)}
    worst = {"fp32": 0.0, "bf16": 0.0}
    nstages = 0

    def run_to(i, face_limit=0, phase_limit=0, first=-1):
        _lib.check(L.hd_set_option(ctx, b"stage_limit_first", first), ctx)          # the limits stop THIS stage only
        _lib.check(L.hd_set_option(ctx, b"face_block_limit", face_limit), ctx)
        _lib.check(L.hd_set_option(ctx, b"xcd_phase_limit", phase_limit), ctx)
        L.hd_debug_limit_ops(ctx, 0, i)
        e.eps(xd, t)

    def check(i, name, what, got, want, stored_bf16):
        rel, mx = _rel(got, want)
        kind = "bf16" if stored_bf16 else "fp32"
        lim = bf16_bound if stored_bf16 else fp32_bound
        worst[kind] = max(worst[kind], rel if rel == rel else 1e9)
        report.append(f"{i:3d} {name:42s} {what:8s} rel {rel:.3e} maxabs {mx:.3e} ({kind} <= {lim:.0e}){'' if rel <= lim else '  <<<<<<'}")

    gated_last = {"denoiser.decoders.0.1": 1, "denoiser.decoders.1.1": 2, "denoiser.decoders.2.1": 3, "denoiser.decoders.3.1": 4}
    enc_blocks = [2, 2, 4, 8]
    try:
        for i, name in enumerate(names):
            parts = name.split(".")
            if parts[0] != "denoiser" or parts[-1] != "conv5" or parts[1] == "middle_blks":
                continue
            if i > 0 and names[i - 1].startswith(".".join(parts[:-1]) + "."):
                continue                                              # a per-GEMM conv5 launch, not a stage
            l, C, H = level_of(name, latent)
            sl = str(l)
            M = B * H * H
            nblk = enc_blocks[l] if parts[1] == "encoders" else 2
            assert int(parts[3]) == nblk - 1, name
            grp = ".".join(parts[:3])
            first = sum(enc_blocks[:l]) if parts[1] == "encoders" else 16 + 8 + 2 * int(parts[2])     # index of the stage's first block
            nstages += 1
            # the launch that produces this stage's input: its own op, or -- folded -- the stage's ENTRY (intro conv at level 0, the down conv of
            # level 0 at level 1): the stage run with NO block (face_block_limit < 0) then leaves the entry's x, which is held against the
            # oracle's arithmetic for that conv on ITS inputs
            producer = ("intro" if l == 0 else f"downs.{l - 1}") if parts[1] == "encoders" else f"ups.{parts[2]}"
            if (names[i - 1] if i > 0 else "") != producer:
                if producer.startswith("downs"):
                    run_to(i)
                    src0 = _nchw(_read(L, ctx, "X" + str(l - 1)), B, C // 2, 2 * H)
                elif producer.startswith("ups"):                      # X holds the encoder's skip, the level above its (HCA) output
                    run_to(i)
                    src0 = _nchw(_read(L, ctx, ("Y" if model.engine.conditional else "X") + str(l + 1)), B, 2 * C, H // 2)
                    skip0 = _nchw(_read(L, ctx, "X" + sl), B, C, H)
                run_to(i + 1, face_limit=-1, first=first)
                cur = _nchw(_read(L, ctx, "X" + sl), B, C, H)
                if producer == "intro":
                    want0 = F.conv2d(x, P["denoiser.intro.weight"], P["denoiser.intro.bias"], padding=1)
                elif producer.startswith("downs"):
                    want0 = O._gemm_conv(src0, P[f"denoiser.{producer}.weight"], P[f"denoiser.{producer}.bias"], PR, stride=2)
                else:
                    want0 = O._up_shuffle(src0, P[f"denoiser.{producer}.0.weight"], 2, PR) + skip0
                check(i, producer + " (stage entry)", "X", _read(L, ctx, "X" + sl)[:M * C], _rows(want0), False)
            else:
                run_to(i)
                cur = _nchw(_read(L, ctx, "X" + sl), B, C, H)
            for b in range(nblk):
                p = f"{grp}.{b}"
                want = O.cond_naf_block(P, p, cur, temb, prec=PR)
                lim = 0 if b == nblk - 1 else b + 1
                run_to(i + 1, face_limit=lim, phase_limit=5 * lim, first=first)
                got = _read(L, ctx, "X" + sl)[:M * C]
                check(i, p + " (stage)", "X", got, _rows(want), False)
                # the block's own contribution x' - x (the residual carries most of the norm of x'): passes through the bf16-stored
                # gate tiles, hence the bf16 bound
                check(i, p + " (stage)", "X'-X", got - _rows(cur), _rows(want - cur), True)
                cur = _nchw(got, B, C, H)
            # exit copies of the whole stage
            if model.engine.conditional and f"{grp}.{nblk - 1}" in gated_last:
                gi = gated_last[f"{grp}.{nblk - 1}"]
                wc = gate_c[gi][:B * C].reshape(B, C, 1, 1)
                ws = gate_s[gi][:M].reshape(B, 1, H, H)
                check(i, name, "Xg", _read(L, ctx, "Xg" + sl)[:M * C], _rows(PR.q(cur * (1.0 + wc + ws))), True)
            else:
                check(i, name, "Xb", _read(L, ctx, "Xb" + sl)[:M * C], _rows(PR.q(cur)), True)
        # the step's last launch when it is the last HCA conv + the ending conv in one (hd_end.hpp): on ITS input, the gated decoder output
        if names[-1] == "ending" and model.engine.conditional and (n < 2 or names[-2] != "hcas.4"):
            run_to(n - 1)
            xg = _nchw(_read(L, ctx, "Xg0"), B, 128, latent)
            run_to(n)
            q = "denoiser.hcas.4"
            y0 = torch.relu(O._conv_bn(xg, P, q + ".fused_mlp.0", q + ".fused_mlp.1", PR, padding=1))
            want = F.conv2d(y0, P["denoiser.ending.weight"], P["denoiser.ending.bias"], padding=1)
            check(n - 1, "hcas.4 + ending (one launch)", "eps", _read(L, ctx, "eps")[:want.numel()], want.reshape(-1), False)
    finally:
        L.hd_set_option(ctx, b"stage_limit_first", -1)
        L.hd_set_option(ctx, b"face_block_limit", 0)
        L.hd_set_option(ctx, b"xcd_phase_limit", 0)
        L.hd_debug_limit_ops(ctx, 0, -1)
    worst["stages"] = nstages
    return worst


def main():
    import argparse
    from hifidiff_amd import synth
    from hifidiff_amd.refiner import FacialRefiner
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--t", type=float, default=500.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "op_forced.txt"))
    ap.add_argument("--stages", action="store_true", help="the persistent stages of the default program, block by block (stage_forced_scan)")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    P = synth.refiner_state_dict(a.latent)
    if not a.stages:
        os.environ["HD_NO_XCD"] = "1"
    m = FacialRefiner(a.latent); m.load_state_dict(P); m.to("cuda")
    x, crl, crf = synth.sample_inputs(a.batch, a.latent)
    report = []
    worst = (stage_forced_scan if a.stages else forced_scan)(m, P, x, crl, crf, a.t, report)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(f"teacher-forced {'stage (block by block)' if a.stages else 'op'} parity, batch {a.batch}, latent {a.latent}, t {a.t}: worst fp32 {worst['fp32']:.3e}, worst bf16-stored {worst['bf16']:.3e}\n")
        f.write("\n".join(report) + "\n")
    bad = [r for r in report if "<<<<<<" in r or "no rule" in r]
    print("\n".join(bad[:40]))
    print(f"worst fp32 {worst['fp32']:.3e}  worst bf16-stored {worst['bf16']:.3e}; {len(bad)} flagged of {len(report)} checks; report in {a.out}")


if __name__ == "__main__":
    main()
