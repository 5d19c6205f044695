"""Which libm the glibc-faithful math mode (OUTFIT_MATH_GLIBC) restates, and whether THIS host runs it.

The mode reproduces the bits of ONE libm build — the one the tables were read from and the restatement was pinned
against (tools/extract_glibc_tables.py, tools/glibc_pin*.c): Ubuntu GLIBC 2.35-0ubuntu3.12, x86-64, the FMA / AVX2
ifunc variants. "Equal to the platform libm" is a statement about that build only: on any other libc (or on a CPU
that selects other ifunc variants) the comparison measures something else, so tests that assert it SKIP there and
bench.py labels its `parity_vs_libm` block instead of asserting. No compute here; nothing under oracle/ is touched.
"""
import ctypes
import ctypes.util
import functools
import hashlib
import os
import platform

PINNED_GLIBC_VERSION = "2.35"
PINNED_LIBM_SHA256 = "e5141752c850ea45691513faadc577133fedf77bcbf19473f97e7247561254b2"
PINNED_LIBM_WHAT = "Ubuntu GLIBC 2.35-0ubuntu3.12, x86-64, Build ID df46fc5774ae8aaaf6efcb97dc7b91532056b898"


def _libm_path():
    """Path of the libm.so.6 MAPPED INTO THIS PROCESS — the object whose code `sin`, `atan2`, ... of a library loaded here
    (the oracle, via ctypes) resolve to. A well-known path says nothing under conda, LD_LIBRARY_PATH or a container
    with its own libm, so the mapping is read from /proc/self/maps after making sure a libm is loaded; only when that
    is not possible (no procfs) the usual locations are tried."""
    try:
        ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")      # map it if nothing has yet
    except OSError:
        pass
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[5].strip() if len(line.split(None, 5)) == 6 else ""
                base = os.path.basename(path)
                if base.startswith("libm.so") or base.startswith("libm-"):
                    return os.path.realpath(path)
    except OSError:
        pass
    for cand in ("/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6",
                 "/usr/lib64/libm.so.6"):
        if os.path.exists(cand):
            return os.path.realpath(cand)
    return None


def _cpu_flags():
    """The flag tokens of the first `flags` line of /proc/cpuinfo (token membership: a flag may be last on its line)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


@functools.lru_cache(maxsize=1)
def host_libm_identity():
    """{'libc', 'version', 'machine', 'fma_avx2', 'libm_path', 'libm_sha256', 'pinned', 'why_not'}."""
    libc, ver = platform.libc_ver()
    out = {"libc": libc or "unknown", "version": ver or "", "machine": platform.machine(), "fma_avx2": False,
           "libm_path": None, "libm_sha256": None, "pinned": False, "why_not": None,
           "pinned_build": PINNED_LIBM_WHAT}
    try:  # the running libc's own answer, when it is glibc
        fn = ctypes.CDLL(None).gnu_get_libc_version
        fn.restype = ctypes.c_char_p
        out["libc"], out["version"] = "glibc", fn().decode()
    except (AttributeError, OSError):
        pass
    out["fma_avx2"] = {"fma", "avx2"} <= _cpu_flags()  # the ifunc variants the restatement follows
    p = _libm_path()
    if p and os.path.isfile(p):
        out["libm_path"] = p
        out["libm_sha256"] = hashlib.sha256(open(p, "rb").read()).hexdigest()
    if out["libc"] != "glibc" or out["machine"] != "x86_64":
        out["why_not"] = "host libm is not glibc on x86-64"
    elif not out["fma_avx2"]:
        out["why_not"] = "CPU without FMA/AVX2: glibc selects other ifunc variants than the restated ones"
    elif out["version"] != PINNED_GLIBC_VERSION:
        out["why_not"] = f"glibc {out['version']}, the restatement follows {PINNED_GLIBC_VERSION}"
    elif out["libm_sha256"] != PINNED_LIBM_SHA256:
        out["why_not"] = (f"libm.so.6 sha256 {str(out['libm_sha256'])[:16]}... is not the build the tables were read "
                          f"from ({PINNED_LIBM_SHA256[:16]}...)")
    else:
        out["pinned"] = True
    return out


def host_has_pinned_libm():
    """True iff `platform libm == what OUTFIT_MATH_GLIBC restates` is a meaningful comparison on this host."""
    return host_libm_identity()["pinned"]


def describe():
    i = host_libm_identity()
    s = f"{i['libc']} {i['version']} {i['machine']} libm sha256 {str(i['libm_sha256'])[:16]}"
    return s + (" (the pinned build)" if i["pinned"] else f" (NOT the pinned build: {i['why_not']})")
