#!/usr/bin/env python3
"""Tables of glibc 2.35's table-driven atan2, tan and acos (IBM Accurate Mathematical Library, sysdeps/ieee754/dbl-64/
e_atan2.c `cij`, s_tan.c `xfg`, asincos.tbl `asncs`, root.tbl `inroot`, powtwo.tbl `powtwo`), read out of the installed
libm.so.6 for the glibc-faithful restatement (oracle/glibcmath2.c, glibcmath3.c, outfit_amd/csrc/device/glibcmath2.hpp, glibcmath3.hpp).

Unlike the sin/cos and exp tables (tools/gen_glibc_tables.py, regenerated with mpmath) these cannot be regenerated:
their nodes x_i are "accurate table" points chosen by a search so that atan(x_i) / tan(x_i) are unusually close to
doubles. They are mathematical constants of a third-party library (glibc, LGPL-2.1) that the reference links against
at run time; what is taken is data (241 x 7 and 186 x 4 doubles), located by the addresses the machine code of
__ieee754_atan2_fma / __tan_fma loads them from, in the exact build named below. The values are sanity-checked
against math.atan / math.tan before anything is written, and the restatement built on them is pinned against the
platform's atan2 / tan bit for bit (tests/test_glibcmath2.py).

Usage: python tools/extract_glibc_tables.py [/lib/x86_64-linux-gnu/libm.so.6]
"""
import hashlib
import math
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBM = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
KNOWN = {  # sha256 of libm.so.6 -> (cij file offset, xfg file offset, description)
    "e5141752c850ea45691513faadc577133fedf77bcbf19473f97e7247561254b2":
        (0xbe0e0, 0xc15c0, "Ubuntu GLIBC 2.35-0ubuntu3.12, x86-64, Build ID df46fc5774ae8aaaf6efcb97dc7b91532056b898"),
}
KNOWN3 = {  # sha256 -> (asncs, inroot, powtwo): the addresses __ieee754_acos_fma (0x77960 in that build) loads them from;
    # then __log_data (e_log_data.c: ln2hi, ln2lo, poly[5], poly1[11], tab[128] = {invc, logc}), read by __log_fma (0x76660)
    "e5141752c850ea45691513faadc577133fedf77bcbf19473f97e7247561254b2": (0xb90a0, 0xb8ca0, 0xb8bc0, 0xb01e0),
}



def license_header(style):
    """The LGPL notice every glibc-derived file carries (tools/glibc_license_header.txt), as a C or C++ comment."""
    txt = open(os.path.join(ROOT, "tools", "glibc_license_header.txt")).read().rstrip("\n").split("\n")
    if style == "c":
        return "/*\n" + "\n".join((" * " + l) if l else " *" for l in txt) + "\n */\n"
    return "\n".join(("// " + l) if l else "//" for l in txt) + "\n//\n"


def main():
    data = open(LIBM, "rb").read()
    sha = hashlib.sha256(data).hexdigest()
    if sha not in KNOWN:
        raise SystemExit(f"{LIBM}: unknown build (sha256 {sha}); the table offsets are known for: "
                         + "; ".join(v[2] for v in KNOWN.values()))
    off_cij, off_xfg, what = KNOWN[sha]
    cij = [struct.unpack("<7d", data[off_cij + 56 * i: off_cij + 56 * i + 56]) for i in range(241)]
    xfg = [struct.unpack("<4d", data[off_xfg + 32 * i: off_xfg + 32 * i + 32]) for i in range(186)]
    # what the tables are: cij[i] = x_i, atan(x_i), then the Taylor coefficients of atan at x_i; x_i ~ (i + 16.25) / 256
    for i, r in enumerate(cij):
        assert abs(r[0] - (i + 16.25) / 256) < 2e-3 and abs(math.atan(r[0]) - r[1]) < 1e-16 and abs(r[2] - 1 / (1 + r[0] ** 2)) < 1e-15, i
    # xfg[i] = x_i, tan(x_i), cot(x_i), low part of cot(x_i); x_i ~ (i + 16) / 256
    for i, r in enumerate(xfg):
        assert abs(r[0] - (i + 16) / 256) < 1e-6 and abs(math.tan(r[0]) - r[1]) < 2e-16 and abs(1 / math.tan(r[0]) - r[2]) < 4e-15, i
    header = (f"glibc 2.35 atan2 `cij` (241 x 7) and tan `xfg` (186 x 4) tables, extracted by tools/extract_glibc_tables.py\n"
              f"from {what} (sha256 {sha[:16]}...). Data of a third-party library (glibc, LGPL-2.1); see the tool.")

    def rows(tab):
        return ",\n".join("  {" + ", ".join(float.hex(v) for v in r) + "}" for r in tab)
    c = (license_header("c") + "/* " + header.replace("\n", "\n * ") + " */\n#pragma once\n"
         f"static const double orc_glibc_cij[241][7] = {{\n{rows(cij)}\n}};\n"
         f"static const double orc_glibc_xfg[186][4] = {{\n{rows(xfg)}\n}};\n")
    open(os.path.join(ROOT, "oracle", "glibc_tables2.h"), "w").write(c)
    d = (license_header("cpp") + "// " + header.replace("\n", "\n// ") + "\n#pragma once\nnamespace outfit {\nnamespace glm {\n"
         f"__device__ const double kCij[241][7] = {{\n{rows(cij)}\n}};\n"
         f"__device__ const double kXfg[186][4] = {{\n{rows(xfg)}\n}};\n}}  // namespace glm\n}}  // namespace outfit\n")
    open(os.path.join(ROOT, "outfit_amd", "csrc", "device", "glibc_tables2.hpp"), "w").write(d)
    print("wrote oracle/glibc_tables2.h and outfit_amd/csrc/device/glibc_tables2.hpp from", what)

    # ---- acos: asncs (2568 doubles: per node x_i, the Taylor coefficients of asin at x_i and asin(x_i) in two parts),
    # inroot (128: 1/sqrt seeds), powtwo (28: powers of two)
    off_a, off_r, off_p, off_l = KNOWN3[sha]
    asncs = struct.unpack("<2568d", data[off_a: off_a + 8 * 2568])
    inroot = struct.unpack("<128d", data[off_r: off_r + 8 * 128])
    powtwo = struct.unpack("<28d", data[off_p: off_p + 8 * 28])
    assert all(powtwo[i] == 2.0 ** i for i in range(28))
    assert abs(inroot[0] - 1.4087) < 1e-3 and abs(inroot[127] - 0.7085) < 1e-3 and all(inroot[i] > inroot[i + 1] for i in range(127))
    for i in range(32):       # 0.125 <= x < 0.25: entries of 11, node, c1, .., asin(node) low / high at +7 / +8
        n = 11 * i
        assert abs(asncs[n + 8] + asncs[n + 7] - math.asin(asncs[n])) < 1e-16 and 0.125 <= asncs[n] < 0.25, i
    for i in range(116, 120):  # 0.953125 <= x < 0.96875: entries of 15, asin(node) at +12
        n = 15 * i + 768
        assert abs(asncs[n + 12] + asncs[n + 11] - math.asin(asncs[n])) < 1e-15 and 0.953 <= asncs[n] < 0.969, i
    logd = struct.unpack("<274d", data[off_l: off_l + 8 * 274])   # 2 + 5 + 11 + 256
    assert logd[0] + logd[1] == math.log(2) and abs(logd[2] + 0.5) < 1e-15 and logd[7] == -0.5
    for i in range(128):
        assert abs(logd[18 + 2 * i + 1] + math.log(logd[18 + 2 * i])) < 1e-15, i      # logc = log(1 / invc)
    header3 = (f"glibc 2.35 acos tables `asncs` (2568), `inroot` (128), `powtwo` (28) and `__log_data` (274), extracted by tools/extract_glibc_tables.py\n"
               f"from {what} (sha256 {sha[:16]}...). Data of a third-party library (glibc, LGPL-2.1); see the tool.")

    def flat(tab, per=4):
        return ",\n".join("  " + ", ".join(float.hex(v) for v in tab[i:i + per]) for i in range(0, len(tab), per))
    c3 = (license_header("c") + "/* " + header3.replace("\n", "\n * ") + " */\n#pragma once\n"
          f"static const double orc_glibc_asncs[2568] = {{\n{flat(asncs)}\n}};\n"
          f"static const double orc_glibc_inroot[128] = {{\n{flat(inroot)}\n}};\n"
          f"static const double orc_glibc_powtwo[28] = {{\n{flat(powtwo)}\n}};\n"
          f"/* ln2hi, ln2lo, A[5], B[11], then 128 x (invc, logc) */\n"
          f"static const double orc_glibc_logdata[274] = {{\n{flat(logd)}\n}};\n")
    open(os.path.join(ROOT, "oracle", "glibc_tables3.h"), "w").write(c3)
    d3 = (license_header("cpp") + "// " + header3.replace("\n", "\n// ") + "\n#pragma once\nnamespace outfit {\nnamespace glm {\n"
          f"__device__ const double kAsncs[2568] = {{\n{flat(asncs)}\n}};\n"
          f"__device__ const double kInroot[128] = {{\n{flat(inroot)}\n}};\n"
          f"// ln2hi, ln2lo, A[5], B[11], then 128 x (invc, logc)\n"
          f"__device__ const double kLogData[274] = {{\n{flat(logd)}\n}};\n}}  // namespace glm\n}}  // namespace outfit\n")
    open(os.path.join(ROOT, "outfit_amd", "csrc", "device", "glibc_tables3.hpp"), "w").write(d3)
    print("wrote oracle/glibc_tables3.h and outfit_amd/csrc/device/glibc_tables3.hpp")


if __name__ == "__main__":
    main()
