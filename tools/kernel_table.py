#!/usr/bin/env python3
"""Per-kernel roofline table of one diffusion step from a rocprofv3 --kernel-trace CSV of bench.py and the library's op
list (HD_DUMP_OPS): algorithmic bytes of every launch (bf16 weights once + activations read and written), its average
duration, GB/s and the fraction of the 8 TB/s HBM peak.  usage: kernel_table.py <kernel_trace.csv> <ops.txt> [batch] [latent]"""
import collections
import csv
import sys

B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = int(sys.argv[4]) if len(sys.argv) > 4 else 16
PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def level_of(name):
    p = name.split(".")
    if p[0] == "denoiser" and p[1] == "encoders":
        return int(p[2])
    if p[0] == "denoiser" and p[1] == "middle_blks":
        return 4
    if p[0] == "denoiser" and p[1] == "decoders":
        return 3 - int(p[2])
    raise ValueError(name)


def dims(l):
    C, H = 128 << l, L >> l
    return C, H * H, B * H * H


def op_bytes(name):
    """(weight bytes, activation bytes) of one launch."""
    p = name.split(".")
    if name == "intro":
        return 128 * 4 * 9 * 4, B * 4 * L * L * 4 + B * L * L * 128 * 6
    if name == "ending":
        return 128 * 4 * 9 * 4, B * L * L * 128 * 4 + B * 4 * L * L * 4 * 3
    if p[0] == "downs":
        l = int(p[1]); C, HW, M = dims(l); C2, _, M2 = dims(l + 1)
        return 4 * C * C2 * 2, M * C * 2 + M2 * C2 * 6
    if p[0] == "ups":
        i = int(p[1]); Ch, _, Mh = dims(4 - i); Cl, _, Ml = dims(3 - i)
        return Ch * 2 * Ch * 2, Mh * Ch * 2 + Ml * Cl * 10
    if p[0] == "hcas":
        i = int(p[1]); C, HW, M = dims(4 - i)
        return (C * C if HW == 1 else 9 * C * C) * 2, M * C * 2 + M * C * 6
    l = level_of(name); C, HW, M = dims(l)
    kind = p[-1]
    if kind == "conv2_gate_pool":
        return 2 * C * C * 2, M * C * 2 + M * C * 2
    if kind == "sca":
        return C * C * 2, B * C * 2 + (2 * M * C * 2 if HW <= 16 else 0)
    if kind == "conv3":
        return C * C * 2, M * C * 2 + M * C * 4 + M * C * 6
    if kind == "conv4":
        return 2 * C * C * 2, M * C * 2 + M * C * 2
    if kind == "conv5":
        if C <= 256:                       # levels 0/1: sca + conv3 + conv4 + conv5 in one launch (hd_chain.hpp)
            return 5 * C * C * 2, M * C * 2 + M * C * 4 + M * C * 6
        return C * C * 2, M * C * 2 + M * C * 4 + M * C * 6
    if kind == "conv1":
        return 2 * C * C * 2, M * C * 2 + M * 2 * C * 4
    if kind == "pool_finish":
        return 0, B * C * 8
    raise ValueError(name)


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
ops = [l.strip() for l in open(sys.argv[2]) if l.strip()]
n = len(ops)
# a diffusion step ends with the ending conv (the intro conv may be folded into the first stage: no launch of its own to look for)
ends = [i for i, r in enumerate(rows) if "ending_conv_kernel" in r["Kernel_Name"]]
starts = [j - n + 1 for i, j in zip(ends, ends[1:]) if j - i == n]
starts = starts[1:] if len(starts) > 2 else starts
avg = [sum(dur[s + k] for s in starts) / len(starts) for k in range(n)]
groups = collections.OrderedDict()


def stage_blocks(k, name):
    """A persistent stage (hd_xcd.hpp at levels 2 / 3, hd_face.hpp at levels 0 / 1) appears as ONE launch named after its last
    block's conv5, and the launch before it does not belong to that block.  Returns the number of blocks it covers (0: an
    ordinary launch)."""
    p = name.split(".")
    if p[0] != "denoiser" or p[-1] != "conv5" or level_of(name) > 3 or L != 16:
        return 0
    if k > 0 and ops[k - 1].startswith(".".join(p[:-1]) + "."):
        return 0
    return int(p[3]) + 1 if p[1] == "encoders" else 2


for k, (name, a) in enumerate(zip(ops, avg)):
    p = name.split(".")
    nb = stage_blocks(k, name)
    if nb:
        key = "L%d %s stage (%d blocks, 1 launch)" % (level_of(name), "enc" if p[1] == "encoders" else "dec", nb)
        C, HW, M = dims(level_of(name))
        # per block: conv1 2C^2 + sca C^2 + conv3 C^2 + conv4 2C^2 + conv5 C^2 = 7 C^2 bf16 weights; activations as the launches it replaces
        w = nb * 7 * C * C * 2
        if C <= 256:                       # conv1, depthwise+gate, then the chain launch (op_bytes of "conv5" at these levels)
            act = nb * sum(op_bytes(".".join(p[:-1]) + "." + q)[1] for q in ("conv1", "conv2_gate_pool", "conv5"))
        else:
            act = nb * sum(op_bytes(".".join(p[:-1]) + "." + q)[1] for q in ("conv2_gate_pool", "sca", "conv3", "conv4"))
            act += nb * (M * C * 2 + M * C * 4 + M * C * 6)
        g = groups.setdefault(key, [0, 0.0, 0, 0])
        g[0] += 1; g[1] += a; g[2] += w; g[3] += act
        continue
    if p[0] == "denoiser":
        key = "L%d %s" % (level_of(name), p[-1]) if level_of(name) < 4 else "mid %s" % p[-1]
    else:
        key = name
    w, act = op_bytes(name)
    if name == "ending" and L == 16 and "hcas.4" not in ops and "hcas.3" in ops:
        # hd_end.hpp: the last HCA conv and the ending conv in one launch -- the HCA weights and its bf16 input, no fp32 round trip of its output
        key = "hcas.4 + ending (one launch)"
        w += 9 * 128 * 128 * 2
        act = B * L * L * 128 * 2 + B * 4 * L * L * 4 * 3
    g = groups.setdefault(key, [0, 0.0, 0, 0])
    g[0] += 1; g[1] += a; g[2] += w; g[3] += act
print("one diffusion step, batch %d, latent %d: %d launches, %.1f us of kernel time (%d steps averaged)" % (B, L, n, sum(avg), len(starts)))
print("%-38s %4s %10s %10s %9s %9s %8s %10s" % ("launch", "n", "weights MB", "activ. MB", "us each", "us total", "GB/s", "% of 8TB/s"))
tw = ta = tt = 0.0
for k, (c, t, w, a) in groups.items():
    gbs = (w + a) / c / (t / c) / 1e3
    print("%-38s %4d %10.2f %10.2f %9.2f %9.1f %8.0f %9.1f%%" % (k, c, w / c / 1e6, a / c / 1e6, t / c, t, gbs, 100 * gbs / PEAK))
    tw += w; ta += a; tt += t
print("%-38s %4d %10.1f %10.1f %9s %9.1f %8.0f %9.1f%%" % ("total", n, tw / 1e6, ta / 1e6, "", tt, (tw + ta) / tt / 1e3, 100 * (tw + ta) / tt / 1e3 / PEAK))
