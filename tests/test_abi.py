"""The C ABI: every function include/*.h declares is exported by the shared libraries, and
the struct layouts agree between the headers (compiled with gcc), the ctypes / numpy
mirrors, the oracle and the reference's documented strides."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

from raytracing_amd import backend, host, types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_librtamd_exports_every_declared_symbol():
    names = _declared("rtamd.h", "rtamd_")
    assert len(names) >= 12
    L = backend.lib()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(backend.EXPORTS)
    assert b"gfx950" in L.rtamd_version()


def test_librtamd_host_exports_every_declared_symbol():
    names = _declared("rtamd_host.h", "rtamd_host_")
    assert len(names) >= 18
    L = host.lib()
    for n in names:
        assert hasattr(L, n), n


def test_restated_reference_producers_are_a_library_of_their_own():
    """The reference's builder, bake and per-frame system restated (tests/producers) are fixture
    infrastructure: declared in their own header, exported by librtamd_testhost.so and by nothing else."""
    text = open(os.path.join(ROOT, "tests", "producers", "rtamd_testhost.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rtamd_host_\w+)\s*\(", text)))
    assert len(names) == 7
    P = host.producers()
    out = subprocess.run(["nm", "-D", "--defined-only", host._LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for n in names:
        assert hasattr(P, n), n
        assert n not in exported, f"{n} is still exported by the product's host library"


def test_struct_layouts_match_reference_strides(oracle):
    # BuildAccelerateStructureSystem.cs:125,131,137,144 and RayTracingPass.cs:83
    assert oracle.struct_sizes() == [116, 24, 12, 180, 144]
    assert [T.NODE.itemsize, T.VERTEX.itemsize, T.INT3.itemsize, T.OBJECT.itemsize, T.FRAME.itemsize] == [116, 24, 12, 180, 144]
    assert T.NODE.fields["Data"][1] == 96 and T.NODE.fields["Flags"][1] == 112
    assert T.OBJECT.fields["albedo"][1] == 4 and T.OBJECT.fields["WorldToLocal"][1] == 52 and T.OBJECT.fields["LocalToWorld"][1] == 116
    assert T.FRAME.fields["CameraInverseProjection"][1] == 64 and T.FRAME.fields["PixelOffset"][1] == 128
    assert T.FRAME.fields["Seed"][1] == 136 and T.FRAME.fields["ReflectionNum"][1] == 140


def test_header_layouts_compiled_with_gcc():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "rtamd_host.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rtamd_node), sizeof(rtamd_vertex), sizeof(rtamd_int3),
         sizeof(rtamd_object_property), sizeof(rtamd_frame_cbuffer), sizeof(rtamd_ext_params), sizeof(rtamd_counters),
         sizeof(rtamd_host_buffers), sizeof(rtamd_host_mesh));
  printf("%zu %zu %zu %zu\n", offsetof(rtamd_ext_params, seed), offsetof(rtamd_ext_params, flags),
         offsetof(rtamd_ext_params, out_t), offsetof(rtamd_ext_params, out_counters));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    sizes = list(map(int, out))
    assert sizes[:5] == [116, 24, 12, 180, 144]
    assert sizes[5] == C.sizeof(T.ExtParams) and sizes[6] == C.sizeof(T.Counters)
    assert sizes[7] == C.sizeof(T.HostBuffers) and sizes[8] == C.sizeof(T.HostMesh)
    assert sizes[9:] == [T.ExtParams.seed.offset, T.ExtParams.flags.offset, T.ExtParams.out_t.offset, T.ExtParams.out_counters.offset]


def test_no_gpu_means_loud_failure():
    """Without a visible GPU context creation must fail with NO_DEVICE (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    ctx = C.c_void_p()
    rc = backend.lib().rtamd_ctx_create(0, C.byref(ctx))
    assert rc == -7 and not ctx.value
    assert b"no CPU fallback" in backend.lib().rtamd_last_error_string(None)


def test_csharp_shim_imports_only_declared_functions_with_the_declared_arity():
    """The P/Invoke declarations a maintainer would drop into the Unity project name functions of
    include/rtamd.h and pass as many arguments as the header declares (the shim cannot be compiled
    here: no C# toolchain)."""
    header = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = {}
    for m in re.finditer(r"\b(rtamd_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        declared[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    shim = open(os.path.join(ROOT, "csharp", "RayTracingPassNative.cs")).read()
    shim = re.sub(r"//[^\n]*", "", shim)
    imports = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern \w+\s+(rtamd_\w+)\s*\(([^;]*?)\)\s*;", shim, flags=re.S)
    assert len(imports) >= 10
    for name, args in imports:
        assert name in declared, name
        n = 0 if not args.strip() else args.count(",") + 1
        assert n == declared[name], (name, n, declared[name])
