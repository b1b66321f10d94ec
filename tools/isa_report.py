#!/usr/bin/env python3
"""ISA lint of the in-tree gfx950 build (CPU only: hipcc cross-compiles, llvm-objdump / llvm-readelf read the result).

For every kernel of the code objects behind hifidiff_amd/csrc/build/*.o:
  * registers / LDS / scratch from the kernel descriptor metadata (llvm-readelf --notes),
  * counts of `scratch_` and `flat_` instructions and of packed-fp32 VALU instructions that take an operand with
    `op_sel:[0,1,0] op_sel_hi:[1,1,0]` (the one operand form profiles/r03_unit_stats_isa/ isolated as the
    non-reproducible one) from the disassembly (llvm-objdump -d).

    python tools/isa_report.py [--out profiles/r04_isa_report.txt]

tests/test_isa_lint.py asserts on the result of collect(); the report file is the per-kernel table.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "hifidiff_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels of the benchmark's 59-launch step that must hold no scratch access
HOT_PREFIXES = ("xcd_stage_kernel", "xcd2_stage_kernel", "naf_face_stage_kernel", "naf_chain_kernel", "naf_strip_dwgate_kernel", "hca_ending_conv_kernel")
BAD_OPSEL = re.compile(r"op_sel:\[0,1,0\]\s+op_sel_hi:\[1,1,0\]")


def demangle(names):
    if not names:
        return {}
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def code_objects(tmp):
    """Extract the gfx950 code object of every object file of the build into tmp; returns their paths."""
    objs = sorted(f for f in os.listdir(BUILD) if f.endswith(".o") and "-hip-" not in f and "-host-" not in f)
    cos = []
    for f in objs:
        dst = os.path.join(tmp, f)
        shutil.copy(os.path.join(BUILD, f), dst)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], check=True, capture_output=True)
        for g in sorted(os.listdir(tmp)):
            if g.startswith(f + ".") and "gfx950" in g:
                cos.append(os.path.join(tmp, g))
    return cos


def kernel_meta(co):
    """name -> dict of the integer fields of the kernel's metadata note."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .agpr_count"):            # first key of a kernel entry
            cur = {"agpr_count": int(v)}
            continue
        if cur is None:
            continue
        if k == "name" and v.startswith("_Z"):
            kernels[v] = cur
        elif k in ("private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count",
                   "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = int(v)
    return kernels


def kernel_isa(co):
    """name -> dict(scratch, flat, bad_opsel, pk_f32, insts) from the disassembly."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), {"scratch": 0, "flat": 0, "bad_opsel": 0, "pk_f32": 0, "insts": 0})
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s:
            continue
        op = s.split()[0]
        cur["insts"] += 1
        if op.startswith("scratch_"):
            cur["scratch"] += 1
        elif op.startswith("flat_"):
            cur["flat"] += 1
        elif op.startswith("v_pk_") and op.endswith("_f32"):
            cur["pk_f32"] += 1
            if BAD_OPSEL.search(s):
                cur["bad_opsel"] += 1
    return out


def waves_per_simd(vgprs):
    alloc = (max(vgprs, 1) + 7) // 8 * 8
    return min(8, 512 // alloc)


def collect():
    """One record per kernel of the build: dict(unit, name, pretty, vgpr, agpr, sgpr, lds, scratch_bytes, vgpr_spill, sgpr_spill,
    scratch, flat, bad_opsel, pk_f32, insts, waves_per_simd)."""
    tmp = tempfile.mkdtemp(prefix="hd_isa_")
    try:
        recs = []
        for co in code_objects(tmp):
            meta, isa = kernel_meta(co), kernel_isa(co)
            pretty = demangle(list(meta))
            for name, m in meta.items():
                i = isa.get(name, {"scratch": 0, "flat": 0, "bad_opsel": 0, "pk_f32": 0, "insts": 0})
                v = m.get("vgpr_count", 0) + m.get("agpr_count", 0)
                recs.append(dict(unit=os.path.basename(co).split(".o.")[0], name=name, pretty=pretty.get(name, name),
                                 vgpr=m.get("vgpr_count", 0), agpr=m.get("agpr_count", 0), sgpr=m.get("sgpr_count", 0),
                                 lds=m.get("group_segment_fixed_size", 0), scratch_bytes=m.get("private_segment_fixed_size", 0),
                                 vgpr_spill=m.get("vgpr_spill_count", 0), sgpr_spill=m.get("sgpr_spill_count", 0),
                                 waves_per_simd=waves_per_simd(v), **i))
        return recs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(pretty, n=110):
    p = re.sub(r"\(.*$", "", pretty).replace("hd::", "").replace("void ", "")
    return p if len(p) <= n else p[: n - 3] + "..."


def is_hot(rec):
    p = rec["pretty"].replace("void ", "").replace("hd::", "")
    return p.startswith(HOT_PREFIXES)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    recs = collect()
    lines = ["ISA report of the in-tree gfx950 build (%d kernels in %d code objects); static LDS only (dynamic LDS is set at launch)"
             % (len(recs), len({r["unit"] for r in recs})),
             "totals: scratch_ instructions %d, flat_ %d, packed-fp32 VALU %d of which op_sel:[0,1,0] op_sel_hi:[1,1,0] %d"
             % (sum(r["scratch"] for r in recs), sum(r["flat"] for r in recs), sum(r["pk_f32"] for r in recs), sum(r["bad_opsel"] for r in recs)),
             "", "%-112s %5s %5s %5s %7s %7s %6s %6s %8s %5s %7s %3s" % ("kernel", "vgpr", "agpr", "sgpr", "lds B", "scr B", "vspil", "sspil", "scratch_", "flat_", "insts", "w/S")]
    for r in sorted(recs, key=lambda r: (not is_hot(r), -r["scratch"], r["pretty"])):
        lines.append("%-112s %5d %5d %5d %7d %7d %6d %6d %8d %5d %7d %3d" % (("* " if is_hot(r) else "  ") + short(r["pretty"]), r["vgpr"], r["agpr"], r["sgpr"], r["lds"],
                                                                          r["scratch_bytes"], r["vgpr_spill"], r["sgpr_spill"], r["scratch"], r["flat"], r["insts"], r["waves_per_simd"]))
    txt = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    sys.stdout.write(txt if not a.out else "\n".join(lines[:40]) + "\n...\n")


if __name__ == "__main__":
    main()
