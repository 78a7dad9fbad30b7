"""Compile the native libraries in-tree.

  raytracing_amd/lib/librtamd.so       HIP kernels + C ABI (hipcc, gfx950 only)
  raytracing_amd/lib/librtamd_host.so  host-side helpers: mesh sources, image codecs, SAH tree writer, camera, scene file (g++, CPU only)
  tests/producers/_build/librtamd_testhost.so   the reference's C# scene producers restated (tests/producers; built only where
                                       that directory exists: fixtures, tests and synthetic bench inputs use it, the package does not)
  oracle/_build/librt_oracle.so        CPU oracle (g++; test infrastructure)

Run as `python -m raytracing_amd.build` or through `__graft_entry__.build()`.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
LIB = os.path.join(ROOT, "raytracing_amd", "lib")
INC = os.path.join(ROOT, "include")

# -ffp-contract=off: the kernels and the oracle must round every product and sum
# separately so that both give the same bits.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]
CXX_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-pthread",
             "-Wall", "-Wno-misleading-indentation"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def device_sources():
    return sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(INC, "rtamd.h")])


def source_sha():
    """Fingerprint of the device library's sources (12 hex digits of a SHA-256 over their names and contents).  It is
    compiled into the library (rtamd_version) and written into every kept hardware-counter record, so that a
    measurement can be told from one of other kernels."""
    import hashlib
    h = hashlib.sha256()
    for p in device_sources():
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    extra = os.environ.get("RTAMD_EXTRA_HIPFLAGS", "")
    if extra:
        h.update(b"flags:" + extra.encode())
    return h.hexdigest()[:12]


def build_device(force=False):
    src = os.path.join(CSRC, "rtamd.hip")
    tlas_src = os.path.join(CSRC, "tlas_build.hip")
    deps = device_sources()
    out = os.path.join(LIB, "librtamd.so")
    if force or _newer(out, deps):
        os.makedirs(LIB, exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        extra = os.environ.get("RTAMD_EXTRA_HIPFLAGS", "").split()      # experiments only
        _run([hipcc] + HIP_FLAGS + extra + ['-DRTAMD_SOURCE_SHA="' + source_sha() + '"', "-I" + INC, "-I" + CSRC, src, tlas_src, "-o", out])
    return out


def build_host(force=False):
    """librtamd_host.so: the package's own host helpers (mesh sources, image codecs, SAH tree writer,
    camera, scene container and scene file: raytracing_amd/csrc/host)."""
    hdir = os.path.join(CSRC, "host")
    srcs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(hdir, "host_internal.h"), os.path.join(CSRC, "wide_bvh.h"), os.path.join(INC, "rtamd.h"),
                   os.path.join(INC, "rtamd_host.h")]
    out = os.path.join(LIB, "librtamd_host.so")
    if force or _newer(out, deps):
        os.makedirs(LIB, exist_ok=True)
        _run([os.environ.get("CXX", "g++")] + CXX_FLAGS + ["-I" + INC, "-I" + hdir] + srcs + ["-o", out, "-lz"])
    return out


PRODUCERS = os.path.join(ROOT, "tests", "producers")
TESTHOST = os.path.join(PRODUCERS, "_build", "librtamd_testhost.so")


def build_testhost(force=False):
    """librtamd_testhost.so: the reference's BVH builder, bake and per-frame flattening restated in C++
    (tests/producers).  Test and fixture infrastructure in a library of its own, which takes the scene
    container, the SAH writer and the error text from librtamd_host.so; skipped where the tests are absent."""
    if not os.path.isdir(PRODUCERS):
        return None
    hdir = os.path.join(CSRC, "host")
    srcs = sorted(os.path.join(PRODUCERS, f) for f in os.listdir(PRODUCERS) if f.endswith(".cpp"))
    host = build_host(force)
    deps = srcs + [host, os.path.join(hdir, "host_internal.h"), os.path.join(INC, "rtamd.h"), os.path.join(INC, "rtamd_host.h")]
    if force or _newer(TESTHOST, deps):
        os.makedirs(os.path.dirname(TESTHOST), exist_ok=True)
        _run([os.environ.get("CXX", "g++")] + CXX_FLAGS + ["-I" + INC, "-I" + hdir] + srcs +
             ["-o", TESTHOST, "-L" + LIB, "-lrtamd_host", "-Wl,-rpath,$ORIGIN/../../../raytracing_amd/lib"])
    return TESTHOST


def build_examples(force=False):
    """examples/render_scene.c, a C99 program over the two headers alone (gcc, linked against both libraries):
    the boundary used from the language a P/Invoke binding mirrors.  Skipped where the examples are absent."""
    src = os.path.join(ROOT, "examples", "render_scene.c")
    if not os.path.isfile(src):
        return None
    exe = os.path.join(ROOT, "examples", "_build", "render_scene")
    deps = [src, build_device(), build_host(), os.path.join(INC, "rtamd.h"), os.path.join(INC, "rtamd_host.h")]
    if force or _newer(exe, deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        _run([os.environ.get("CC", "gcc"), "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + INC, src, "-o", exe,
              "-L" + LIB, "-lrtamd", "-lrtamd_host", "-Wl,-rpath,$ORIGIN/../../raytracing_amd/lib"])
    return exe


def build_oracle(force=False):
    """The checker, twice: the declared arithmetic (every `/` of the shader a true division), and the
    reciprocal-multiplication reading (used only to count what the choice between the two moves)."""
    odir = os.path.join(ROOT, "oracle")
    src = os.path.join(odir, "rt_oracle.cpp")
    out = os.path.join(odir, "_build", "librt_oracle.so")
    lit = os.path.join(odir, "_build", "librt_oracle_reciprocal.so")
    for target, extra in ((out, []), (lit, ["-DRT_ORACLE_RECIPROCAL_DIVISIONS"])):
        if force or _newer(target, [src]):
            os.makedirs(os.path.dirname(target), exist_ok=True)
            _run([os.environ.get("CXX", "g++")] + CXX_FLAGS + extra + [src, "-o", target])
    return out


def build_all(force=False):
    return build_device(force), build_host(force), build_testhost(force), build_oracle(force), build_examples(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
