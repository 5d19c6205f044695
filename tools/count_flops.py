#!/usr/bin/env python3
"""Counted fp64 operations per object of the Gauss-IOD path (SURVEY §8d asks for a counted figure in place of its
~1e7 estimate): the CPU restatement (oracle/) is built with coverage instrumentation, run on the 1 000-object
fixture (BASELINE configs[1], bench parameters), and every executed source line is weighted with the scalar fp64
arithmetic instructions the compiler emitted for it.

  flops(line) = sum over the line's machine instructions of {add, sub, mul, div, sqrt: 1; fma: 2; compare/min/max: 1}
  F_obj       = sum over lines of executions(line) x flops(line) / objects        [libm calls counted separately]

Build: gcc -O1 -g --coverage -ffp-contract=off (no vectorisation, one basic block per simple statement; lines that
hold several blocks — `for` headers — are attributed with the line's execution count, a few-percent effect).
Two work decompositions are counted: `reference` (Earth re-interpolated per candidate and observation, as
observation_ephemeris.rs:303-318 does) and `hoisted` (once per observation — what the device computes, minus the
iterations its exact fixed-point exit skips). Test infrastructure: loads the oracle, never the product.

Usage: python tools/count_flops.py [n_objects]  -> writes profiles/flops_per_object.json
"""
import collections
import ctypes as C
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
ORACLE = os.path.join(ROOT, "oracle")
SRC = ["crmath.c", "glibcmath.c", "kepler.c", "orbit.c", "aberth.c", "gauss.c", "triplets.c", "ephem.c", "trajectory.c",
       "diffcor.c", "earthrot.c"]
ARITH = {"addsd": 1, "subsd": 1, "mulsd": 1, "divsd": 1, "sqrtsd": 1, "maxsd": 1, "minsd": 1, "comisd": 1, "ucomisd": 1,
         "vaddsd": 1, "vsubsd": 1, "vmulsd": 1, "vdivsd": 1, "vsqrtsd": 1, "addpd": 2, "subpd": 2, "mulpd": 2, "divpd": 2}
# nominal fp64 operations of one libm call (glibc 2.35 dbl-64 routines, typical path; fdlibm-style counts)
LIBM_FLOPS = {"sin": 40, "cos": 40, "sincos": 80, "acos": 70, "atan2": 80, "sinh": 45, "cosh": 45, "log": 35, "exp": 30,
              "tan": 70, "hypot": 15, "fmod": 15, "cbrt": 35, "fma": 2}
LIBM = ("sin", "cos", "sincos", "acos", "atan2", "sinh", "cosh", "log", "exp", "tan", "hypot", "fmod", "cbrt", "fma"
        )  # (the binary128 *q functions share source lines with the libm calls but never run with orc_use_libm = 1)


def build(tmp):
    objs = []
    for s in SRC:
        o = os.path.join(tmp, s.replace(".c", ".o"))
        subprocess.run(["gcc", "-O1", "-g", "--coverage", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math",
                        "-D_GNU_SOURCE", "-I" + ORACLE, "-c", os.path.join(ORACLE, s), "-o", o], check=True, cwd=tmp)
        objs.append(o)
    so = os.path.join(tmp, "liboracle_cov.so")
    subprocess.run(["gcc", "-shared", "--coverage", "-o", so] + objs + ["-lquadmath", "-lm", "-lpthread"], check=True)
    return so, objs


def static_flops(objs):
    """{(file, line): (flops, {libm function: calls})} from the objects' disassembly with line info."""
    per_line = collections.defaultdict(lambda: [0, collections.Counter()])
    for o in objs:
        out = subprocess.run(["objdump", "-dl", "--no-show-raw-insn", "-r", o], capture_output=True, text=True, check=True).stdout
        cur = None
        pending_call = None
        for ln in out.splitlines():
            m = re.match(r"^(/[^:]+\.[ch]):(\d+)", ln)
            if m:
                cur = (os.path.basename(m.group(1)), int(m.group(2)))
                continue
            m = re.match(r"^\s+[0-9a-f]+: R_X86_64_(?:PLT32|PC32)\s+(\w+)", ln)
            if m:
                if pending_call and m.group(1) in LIBM:
                    per_line[pending_call][1][m.group(1)] += 1
                pending_call = None
                continue
            m = re.match(r"^\s+[0-9a-f]+:\s+(\S+)(.*)$", ln)
            if m and cur:
                op = m.group(1)
                if op in ARITH:
                    per_line[cur][0] += ARITH[op]
                elif op.startswith("vfmadd") or op.startswith("vfmsub") or op.startswith("vfnmadd"):
                    per_line[cur][0] += 2
                pending_call = cur if op in ("call", "jmp") else None
                continue
    return per_line


def line_counts(tmp):
    subprocess.run(["gcov", "-o", tmp] + [os.path.join(ORACLE, s) for s in SRC], cwd=tmp, capture_output=True, check=True)
    counts = {}
    for g in glob.glob(os.path.join(tmp, "*.gcov")):
        name = os.path.basename(g)[:-5]
        for ln in open(g, errors="replace"):
            m = re.match(r"^\s*([0-9]+)\*?:\s*(\d+):", ln)
            if m:
                counts[(name, int(m.group(2)))] = int(m.group(1))
    return counts


def run_case(so, n, hoist):
    code = f"""
import sys, ctypes as C
sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'tests')!r}]
import numpy as np
from outfit_amd import synth, _abi
import outfit_amd
L = C.CDLL({so!r})
ds = synth.make_dataset({n})
batch = outfit_amd.ObsBatch.from_dict(ds); eph = outfit_amd.Ephemeris.from_dict(ds['ephem'])
p = synth.bench_params(outfit_amd.IODParams()).as_c()
out = (_abi.GaussResultC * {n})()
C.c_int.in_dll(L, 'orc_use_libm').value = 1
bc, ec = batch.as_c(), eph.as_c()
rc = L.orc_fit_full_iod(C.byref(p), C.byref(bc), C.byref(ec), 1, {hoist}, out)
assert rc == 0
print(sum(1 for r in out if r.status == 0))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    return int(r.stdout.strip())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    result = {"objects": n, "workload": "BASELINE configs[1] shape: 20 obs, 100 triplets per object, bench parameters",
              "method": "gcov line executions x scalar fp64 arithmetic instructions per line (gcc -O1, no contraction); fma = 2"}
    for label, hoist in (("reference", 0), ("hoisted", 1)):
        tmp = tempfile.mkdtemp(prefix="flops_")
        try:
            so, objs = build(tmp)
            ok = run_case(so, n, hoist)
            stat = static_flops(objs)
            cnt = line_counts(tmp)
            flops = 0
            libm = collections.Counter()
            per_file = collections.Counter()
            for key, (f, calls) in stat.items():
                c = cnt.get(key, 0)
                flops += c * f
                per_file[key[0]] += c * f
                for k, v in calls.items():
                    libm[k] += c * v
            libm_flops = sum(LIBM_FLOPS.get(k, 50) * v for k, v in libm.items())
            result[label] = {"orbits": ok, "fp64_ops_per_object": flops / n,
                             "libm_fp64_ops_per_object_nominal": libm_flops / n,
                             "total_fp64_ops_per_object": (flops + libm_flops) / n,
                             "libm_calls_per_object": {k: v / n for k, v in sorted(libm.items()) if v},
                             "fp64_ops_per_object_by_file": {k: v / n for k, v in per_file.most_common()}}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "flops_per_object.json"), "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
