"""The readers of untrusted files (sky images: OpenEXR incl. PIZ and Radiance RGBE; COLLADA meshes; the flattened
scene file) under a mutation fuzzer built with AddressSanitizer and UBSan (tests/fuzz/*.cpp compiled together with
the host sources): a damaged file must come back as an error code or as a well-formed result, never as a memory
error, an uncaught exception or an allocation sized by a number in the file."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_images as ti  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DAE = """<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="meter" meter="1"/><up_axis>Z_UP</up_axis></asset>
  <library_geometries>
    <geometry id="g" name="g">
      <mesh>
        <source id="g-pos">
          <float_array id="g-pos-array" count="12">0 0 0 1 0 0 0 1 0 0 0 1</float_array>
          <technique_common><accessor source="#g-pos-array" count="4" stride="3"><param name="X" type="float"/><param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common>
        </source>
        <source id="g-nrm">
          <float_array id="g-nrm-array" count="12">0 0 -1 0 -1 0 -1 0 0 0.577 0.577 0.577</float_array>
          <technique_common><accessor source="#g-nrm-array" count="4" stride="3"><param name="X" type="float"/><param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common>
        </source>
        <vertices id="g-vtx"><input semantic="POSITION" source="#g-pos"/></vertices>
        <triangles material="m" count="4">
          <input semantic="VERTEX" source="#g-vtx" offset="0"/>
          <input semantic="NORMAL" source="#g-nrm" offset="1"/>
          <p>0 0 2 0 1 0 0 1 1 1 3 1 0 2 3 2 2 2 1 3 2 3 3 3</p>
        </triangles>
      </mesh>
    </geometry>
  </library_geometries>
</COLLADA>
"""


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("fuzz"), "fuzz_readers")


def _build(tmp_path, name):
    cxx = os.environ.get("CXX", "g++")
    if not shutil.which(cxx):
        pytest.skip("no C++ compiler")
    hdir = os.path.join(ROOT, "raytracing_amd", "csrc", "host")
    pdir = os.path.join(ROOT, "tests", "producers")
    srcs = [os.path.join(ROOT, "tests", "fuzz", name + ".cpp")]
    srcs += sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".cpp"))
    srcs += sorted(os.path.join(pdir, f) for f in os.listdir(pdir) if f.endswith(".cpp"))
    exe = str(tmp_path / name)
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + hdir] + srcs + ["-o", exe, "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this compiler has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def _run(exe, iterations, work, seeds):
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe, str(iterations), "7", str(work)] + [str(s) for s in seeds], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    read, refused = [int(x) for x in r.stdout.replace(":", "").split() if x.isdigit()][1:3]
    assert read > 0 and refused > 0, r.stdout        # the edits neither all bounce off nor all pass unnoticed
    return r.stdout


def test_image_readers_survive_damaged_files(tmp_path, fuzzer):
    exe = fuzzer
    seeds = []
    k = 0
    for compression in (0, 1, 2, 3, 4):
        for pixel_type in (1, 2):
            p = tmp_path / f"seed{k}.exr"
            ti.write_exr(p, ti._picture(37, 29, 7 + k).astype(np.float32), pixel_type, compression)
            seeds.append(p)
            k += 1
    p = tmp_path / "flat.exr"
    ti.write_exr(p, ti._picture(40, 50, 3).astype(np.float32), 1, 4, piz_flat_code=True)
    seeds.append(p)
    q = ti._rgbe(ti._picture(11, 40, 7)[..., :3])
    for rle in (0, 1):
        body = b"".join(ti._hdr_rle_scanline(q[y]) if rle else q[y].tobytes() for y in range(q.shape[0]))
        p = tmp_path / f"seed{rle}.hdr"
        p.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 11 +X 40\n" + body)
        seeds.append(p)
    _run(exe, 6000, tmp_path / "work", seeds)


def test_mesh_and_scene_file_readers_survive_damaged_files(tmp_path, fuzzer):
    from raytracing_amd import scenes
    exe = fuzzer
    dae = tmp_path / "seed.dae"
    dae.write_text(DAE)
    scene = tmp_path / "seed.rtscene"
    scenes.few_spheres(64, 36)[0].save_file(str(scene))
    _run(exe, 6000, tmp_path / "work", [dae, scene])


def test_known_bad_files_are_refused(tmp_path):
    """The cases the fuzzer found first, kept as plain tests."""
    import struct
    from raytracing_amd import host as H, scenes
    # an <input> without a source reference used to throw out of the C entry point
    bad = tmp_path / "nosource.dae"
    bad.write_text(DAE.replace('source="#g-pos"', 'source=""'))
    with pytest.raises(Exception):
        H.Mesh.load_dae(str(bad))
    # a mesh whose <p> is cut off
    cut = tmp_path / "cut.dae"
    cut.write_text(DAE[:DAE.index("<p>") + 5])
    with pytest.raises(Exception):
        H.Mesh.load_dae(str(cut))
    # a scene file whose header claims a billion nodes: refused by its size, not by running out of memory
    good = tmp_path / "good.rtscene"
    scenes.few_spheres(64, 36)[0].save_file(str(good))
    raw = bytearray(good.read_bytes())
    raw[8:12] = struct.pack("<i", (1 << 30) - 1)
    huge = tmp_path / "huge.rtscene"
    huge.write_bytes(bytes(raw))
    with pytest.raises(Exception):
        H.SceneBuffers.load_file(str(huge))
    # and a good file still loads
    assert len(H.SceneBuffers.load_file(str(good)).triangles) > 0
