#!/usr/bin/env python3
"""Static (and, with execution counts, dynamic) opcode account of one kernel of outfit_amd/csrc/kernels.hip.

    python tools/isa_account.py [--kernel k_gauss_correct_fastILb0E] [--defs "-DX"] [--asm /tmp/k.s] [--blocks]
                                [--weights weights.json]

Compiles the device code to gfx950 assembly (`hipcc --cuda-device-only -S`, the flags of the Makefile), cuts out the
kernel, splits it into basic blocks (the compiler's .LBB labels) and classes every instruction:
  f64     v_add/mul/fma/fmac_f64             (what SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 count)
  trans   v_rcp/rsq/sqrt_f64                 (SQ_INSTS_VALU_TRANS_F64; 16 issue cycles instead of 4)
  div*    v_div_scale/fmas/fixup_f64         (the scaffolding of a full IEEE division: 4 per division)
  cmp, cndmask, mov (v_mov / v_accvgpr), lane (v_readlane / v_writelane / v_readfirstlane: spilled SGPRs), v_other
  salu, smem, lds, vmem:*, wait
`--weights`: {"<block label or region name>": executions per wave, ...} (from a stats variant: trips, Newton rounds,
series term passes, refills) turns the static histogram into a dynamic one; regions are given in the JSON as
{"regions": {"name": ["LBB23_56", ...]}, "executions": {"name": n}}.
VERDICT r4 item 3 asked for this account; DESIGN.md (K2b) quotes its output.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU = ("f64", "trans", "div*", "cmp", "cndmask", "mov", "lane", "v_other")


def cls(op):
    if op.startswith("v_"):
        if re.match(r"v_(add|mul|fma|fmac)_f64", op):
            return "f64"
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            return "trans"
        if re.match(r"v_div_(scale|fmas|fixup)_f64", op):
            return "div*"
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "mov"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "lane"
        return "v_other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem:" + op.split("_")[0]
    return "other"


def compile_asm(defs, out):
    src = os.path.join(ROOT, "outfit_amd", "csrc", "kernels.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
           "--cuda-device-only", "-S", src, "-o", out] + defs.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(src))


def kernel_body(asm, needle):
    lines = open(asm).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", l)]
    if not start:
        raise SystemExit(f"no kernel matching {needle}")
    i0 = start[0]
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[i0].split(":")[0], lines[i0 + 1:i1]


def blocks_of(body):
    blocks, cur = [], ("entry", [])
    for l in body:
        m = re.match(r"^\.(LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur[1].append(t.split()[0])
    blocks.append(cur)
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_gauss_correct_fastILb0E")
    ap.add_argument("--defs", default="")
    ap.add_argument("--asm", default="/tmp/isa_account.s")
    ap.add_argument("--no-compile", action="store_true")
    ap.add_argument("--blocks", action="store_true", help="print every block with >= 12 VALU instructions")
    ap.add_argument("--weights")
    a = ap.parse_args()
    if not a.no_compile:
        compile_asm(a.defs, a.asm)
    name, body = kernel_body(a.asm, a.kernel)
    blocks = blocks_of(body)
    hist = {n: collections.Counter(cls(o) for o in ops) for n, ops in blocks}
    total = collections.Counter()
    for c in hist.values():
        total.update(c)
    nv = sum(total[k] for k in VALU)
    print(f"# {name}: {sum(total.values())} instructions, {nv} VALU, fp64 share of VALU (add/mul/fma/trans) "
          f"{(total['f64'] + total['trans']) / nv:.3f}, full divisions {total['div*'] // 4}")
    print("static: " + " ".join(f"{k}={v}" for k, v in sorted(total.items())))
    if a.blocks:
        for n, ops in blocks:
            c = hist[n]
            v = sum(c[k] for k in VALU)
            if v >= 12:
                print(f"  {n:10s} n={len(ops):4d} valu={v:4d} " + " ".join(f"{k}={c[k]}" for k in sorted(c)))
    if a.weights:
        w = json.load(open(a.weights))
        dyn = collections.Counter()
        per_region = {}
        for region, labels in w["regions"].items():
            c = collections.Counter()
            for lab in labels:
                c.update(hist[lab])
            n = w["executions"][region]
            per_region[region] = (n, c)
            for k, v in c.items():
                dyn[k] += n * v
        dv = sum(dyn[k] for k in VALU)
        print(f"dynamic (per wave): {dv:.0f} VALU, fp64 share {(dyn['f64'] + dyn['trans']) / dv:.3f}")
        print("  " + " ".join(f"{k}={dyn[k] / dv:.3f}" for k in VALU))
        for region, (n, c) in per_region.items():
            v = sum(c[k] for k in VALU)
            print(f"  {region:24s} x{n:9.1f}  valu/exec={v:4d}  share={n * v / dv:.3f}  " +
                  " ".join(f"{k}={c[k]}" for k in VALU if c[k]))


if __name__ == "__main__":
    sys.exit(main())
