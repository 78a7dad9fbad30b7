"""Regenerates the committed golden vectors with the CPU oracle.

    python tests/golden/make_golden.py

The reference holds no golden images or known-answer vectors and cannot be run here, so
these files do not pin the oracle to the reference (PARITY UNPINNED); they pin the oracle
and the scene producers against unintended change, and give the GPU tests fixed expected
outputs.  Inputs are the seeded scenes of raytracing_amd.scenes; outputs are per pixel
(object, triangle), t, RGBA for REF mode and RGBA for PT mode, plus the work counters.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from raytracing_amd import host as H, scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "few_spheres_64x36": lambda: (scenes.few_spheres(64, 36), 64, 36),
    "random_spheres_g6_96x54": lambda: (scenes.random_spheres(96, 54, grid=6), 96, 54),
    "mesh_20k_96x54": lambda: (scenes.mesh_scene(96, 54, target_tris=20000, procedural=True), 96, 54),
}
PT = dict(spp=4, max_bounces=4, seed=1234)


REFERENCE_MODELS = "/root/reference/Assets/Models"


def reference_fixtures(summary):
    """Fixtures derived from the reference's own assets (only where its checkout exists): the three
    meshes as the COLLADA loader reads them (arrays, not the .dae text), the flattened buffers of the
    reference's sample scene with its sample camera, and the oracle's renders of it."""
    if not os.path.exists(os.path.join(REFERENCE_MODELS, "peter.dae")):
        print("no reference checkout: reference-derived fixtures left as they are")
        return
    arrays = {}
    for name in ("peter", "bunny", "cow"):
        m = H.Mesh.load_dae(os.path.join(REFERENCE_MODELS, name + ".dae"))
        arrays[name + "_positions"], arrays[name + "_normals"], arrays[name + "_indices"] = m.positions, m.normals, m.indices
    np.savez_compressed(scenes.MODELS_NPZ, **arrays)
    # BVH node dump of the bunny (4 968 triangles), reference median builder
    bunny = scenes.load_model("bunny", REFERENCE_MODELS)
    aabbs, pts = bunny.triangle_aabbs()
    np.save(os.path.join(HERE, "bvh_bunny.npy"), H.BoundingVolumeHierarchy.Build(pts, aabbs))
    # the sample scene, 640x360
    w, h = 640, 360
    buf, frame, sky = scenes.reference_scene(REFERENCE_MODELS, w, h)
    buf.save_file(os.path.join(HERE, "reference_scene.rtscene"))
    ref = O.render(buf, frame, w, h, sky=sky, mode=0, nthreads=8)
    pw, ph = 320, 180
    _, pframe, _ = scenes.reference_scene(REFERENCE_MODELS, pw, ph)
    pt = O.render(buf, pframe, pw, ph, sky=sky, mode=1, want_ids=False, want_t=False, nthreads=8, **PT)
    np.savez_compressed(os.path.join(HERE, "reference_scene_640x360.npz"), frame=frame, ids=ref["ids"], t=ref["t"],
                        ref_rgba=ref["rgba"], pt_frame=pframe, pt_rgba=pt["rgba"])
    summary["reference_scene_640x360"] = dict(width=w, height=h, ref_counters=ref["counters"], pt_counters=pt["counters"], pt=PT,
                                              pt_size=[pw, ph], n_tlas=len(buf.tlas), n_blas_nodes=len(buf.blas),
                                              n_triangles=len(buf.triangles), n_objects=len(buf.objects),
                                              covered=float((ref["ids"][..., 0] >= 0).mean()))


def main():
    summary = {}
    for name, make in CASES.items():
        (buf, frame, sky), w, h = make()
        ref = O.render(buf, frame, w, h, sky=sky, mode=0)
        pt = O.render(buf, frame, w, h, sky=sky, mode=1, want_ids=False, want_t=False, **PT)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=ref["ids"], t=ref["t"], ref_rgba=ref["rgba"],
                            pt_rgba=pt["rgba"])
        summary[name] = dict(width=w, height=h, ref_counters=ref["counters"], pt_counters=pt["counters"], pt=PT,
                             n_tlas=len(buf.tlas), n_blas_nodes=len(buf.blas), n_triangles=len(buf.triangles),
                             n_objects=len(buf.objects),
                             blas_crc=int(np.frombuffer(buf.blas.tobytes(), dtype=np.uint32).sum(dtype=np.uint64)),
                             tlas_crc=int(np.frombuffer(buf.tlas.tobytes(), dtype=np.uint32).sum(dtype=np.uint64)))
    # BVH node dumps of the two smallest meshes
    for mname, mesh in (("quad", H.Mesh.quad()), ("cube", H.Mesh.cube())):
        aabbs, pts = mesh.triangle_aabbs()
        nodes = H.BoundingVolumeHierarchy.Build(pts, aabbs)
        np.save(os.path.join(HERE, f"bvh_{mname}.npy"), nodes)
    reference_fixtures(summary)
    json.dump(summary, open(os.path.join(HERE, "summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
