"""Seeded, Unity-free scene definitions for the configurations in BASELINE.json.

Every function returns (SceneBuffers, FRAME record, sky (h,w,4) float32).
Material values come from the reference's prefabs / materials where they exist:
  metal    specular (0.8,0.8,0.8)      Assets/Prefabs/Sphere.prefab:113-115
  gold     specular (1,0.78,0.34)      Assets/Prefabs/bunny.001.prefab:100
  diffuse  albedo 1, specular 0.1      Assets/Prefabs/Wall.prefab:114-115
  light    emission (1,1,1)            Assets/Material/Wall2.mat:133
"""
import os

import numpy as np

from . import host as H
from . import types as T


def procedural_sky(w=256, h=128):
    """Gradient sky standing in for the missing .exr files: horizon-to-zenith blue/white,
    a warm band near the sun azimuth, dark ground. Values exceed 1 near the sun (HDR)."""
    v = (np.arange(h, dtype=np.float32) + 0.5) / h          # 0 = +Y pole after the shader's -theta mapping wraps
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    V, U = np.meshgrid(v, u, indexing="ij")
    up = np.cos(np.pi * V).astype(np.float32)                 # +1 zenith .. -1 nadir
    t = np.clip(up * 0.5 + 0.5, 0, 1).astype(np.float32)
    sky = np.zeros((h, w, 4), dtype=np.float32)
    sky[..., 0] = 0.9 - 0.6 * t
    sky[..., 1] = 0.95 - 0.4 * t
    sky[..., 2] = 1.0 - 0.05 * t
    sun = np.exp(-((U - 0.3) ** 2) / 0.002 - ((V - 0.35) ** 2) / 0.004).astype(np.float32)
    sky[..., 0] += 6.0 * sun
    sky[..., 1] += 5.0 * sun
    sky[..., 2] += 3.5 * sun
    ground = up < 0
    sky[ground, 0:3] *= np.float32(0.25)
    sky[..., 3] = 1.0
    return sky


def _rot_x(deg):
    return H.quat_axis_angle((1, 0, 0), deg)


def look_at_camera(eye, target, fov, width, height):
    fwd = np.asarray(target, dtype=np.float64) - np.asarray(eye, dtype=np.float64)
    q = H.quat_look_rotation(fwd)
    return H.camera(eye, q, fov, width / height)


def few_spheres(width=640, height=360):
    """Config 1: three instanced UV spheres (one BLAS, 960 triangles), a floor quad and one
    emissive quad."""
    sys = H.BuildAccelerateStructureSystem()
    sphere = sys.BakeBLAS(H.Mesh.uv_sphere(32, 16, 0.5))
    quad = sys.BakeBLAS(H.Mesh.quad())
    sys.AddEntity(sphere, H.trs((-1.3, 0.5, 0.0), s=(1, 1, 1)), albedo=(0, 0, 0), specular=(0.8, 0.8, 0.8))
    sys.AddEntity(sphere, H.trs((0.0, 0.6, 0.4), s=(1.2, 1.2, 1.2)), albedo=(0, 0, 0), specular=(1.0, 0.78, 0.34))
    sys.AddEntity(sphere, H.trs((1.3, 0.4, -0.3), s=(0.8, 0.8, 0.8)), albedo=(1, 1, 1), specular=(0.1, 0.1, 0.1))
    sys.AddEntity(quad, H.trs((0, 0, 0), _rot_x(90), (100, 100, 1)), albedo=(1, 1, 1), specular=(0.1, 0.1, 0.1))
    sys.AddEntity(quad, H.trs((0, 3.0, 0.5), _rot_x(-90), (2, 2, 1)), albedo=(0, 0, 0), specular=(0, 0, 0), emission=(1, 1, 1))
    buf = sys.OnUpdate()
    sys.close()
    frame = look_at_camera((0.0, 1.6, -5.5), (0.0, 0.5, 0.0), 40.0, width, height)
    return buf, frame, procedural_sky()


def random_spheres(width=1920, height=1080, grid=22, tree=H.BuildAccelerateStructureSystem.MEDIAN):
    """Config 2: floor + grid*grid jittered sphere instances of ONE BLAS (484 at grid=22),
    radius 0.2-0.5; 80 % Lambert, 15 % metal, 5 % emissive."""
    sys = H.BuildAccelerateStructureSystem()
    sphere = sys.BakeBLAS(H.Mesh.uv_sphere(32, 16, 0.5), tree)
    quad = sys.BakeBLAS(H.Mesh.quad(), tree)
    sys.AddEntity(quad, H.trs((0, 0, 0), _rot_x(90), (100, 100, 1)), albedo=(0.8, 0.8, 0.8), specular=(0.04, 0.04, 0.04))
    rp = np.random.RandomState(1)
    rm = np.random.RandomState(2)
    spacing = 1.2
    for iz in range(grid):
        for ix in range(grid):
            r = float(rp.uniform(0.2, 0.5))
            x = (ix - (grid - 1) / 2) * spacing + float(rp.uniform(-0.3, 0.3))
            z = (iz - (grid - 1) / 2) * spacing + float(rp.uniform(-0.3, 0.3))
            kind = float(rm.uniform())
            col = rm.uniform(0.2, 0.9, size=3)
            if kind < 0.80:
                mat = dict(albedo=tuple(col), specular=(0.04, 0.04, 0.04))
            elif kind < 0.95:
                mat = dict(albedo=(0, 0, 0), specular=tuple(0.6 + 0.4 * (col - 0.2) / 0.7))
            else:
                mat = dict(albedo=(0, 0, 0), specular=(0, 0, 0), emission=tuple(4.0 * col))
            sys.AddEntity(sphere, H.trs((x, r, z), s=(2 * r, 2 * r, 2 * r)), **mat)
    buf = sys.OnUpdate()
    sys.close()
    frame = look_at_camera((0.0, 7.0, -22.0), (0.0, 0.3, -2.0), 38.0, width, height)
    return buf, frame, procedural_sky()


def _bumpy_sphere(segments, rings, seed):
    """Procedural stand-in for a scanned model: a sphere displaced by a few sine lobes."""
    m = H.Mesh.uv_sphere(segments, rings, 0.5)
    rs = np.random.RandomState(seed)
    p = m.positions.astype(np.float64)
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    disp = np.zeros(len(p))
    for _ in range(6):
        k = rs.uniform(2.0, 9.0, size=3)
        ph = rs.uniform(0, 2 * np.pi)
        disp += rs.uniform(0.01, 0.05) * np.sin(n @ k + ph)
    p = n * (0.5 + disp)[:, None]
    # normals from the displaced surface: area-weighted face normals
    tri = m.indices
    fn = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, tri[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    vn = np.where(ln > 0, vn / np.maximum(ln, 1e-30), n)
    return H.Mesh(p.astype(np.float32), vn.astype(np.float32), tri)


MODELS_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "reference_models.npz")


def load_model(name, models_dir=None):
    """One of the reference's meshes (peter, bunny, cow: Assets/Models/*.dae) as the COLLADA loader
    reads it: from the .dae when `models_dir` holds it, otherwise from the committed arrays the
    loader produced (raytracing_amd/data/reference_models.npz, written by tests/golden/make_golden.py)."""
    if models_dir and os.path.exists(os.path.join(models_dir, name + ".dae")):
        return H.Mesh.load_dae(os.path.join(models_dir, name + ".dae"))
    z = np.load(MODELS_NPZ, allow_pickle=False)
    return H.Mesh(z[name + "_positions"], z[name + "_normals"], z[name + "_indices"])


def have_models(models_dir=None):
    return bool(models_dir and os.path.exists(os.path.join(models_dir, "peter.dae"))) or os.path.exists(MODELS_NPZ)


def mesh_scene(width=1920, height=1080, models_dir=None, target_tris=1_000_000,
               tree=H.BuildAccelerateStructureSystem.MEDIAN, procedural=False):
    """Config 3: three unique large meshes (~1 M triangles in total) + three emissive walls.

    The meshes are the reference's peter / bunny / cow (Assets/Models/peter.dae:56,78,
    bunny.dae:42,64, cow.dae:42,64) subdivided 2 / 3 / 2 times = 1 051 936 triangles (1 / 2 / 1
    rounds = 165 256 and no subdivision = 50 842 for smaller targets); with `procedural` (or
    when neither the .dae files nor their committed arrays are there), three displaced spheres
    with the same triangle budget split (61 % / 30 % / 9 %)."""
    meshes = []
    if not procedural and have_models(models_dir):
        scale = target_tris / 1_051_936.0
        rounds = (2, 3, 2) if scale > 0.5 else ((1, 2, 1) if scale > 0.1 else (0, 0, 0))
        for name, r in zip(("peter", "bunny", "cow"), rounds):
            meshes.append(load_model(name, models_dir).subdivide(r))
    else:
        for frac, seed in ((0.61, 11), (0.30, 12), (0.09, 13)):
            tris = max(64, int(target_tris * frac))
            rings = max(4, int(round(np.sqrt(tris / 4.0))))      # 2*seg*(rings-1) ~ tris with seg = 2*rings
            meshes.append(_bumpy_sphere(2 * rings, rings, seed))
    sys = H.BuildAccelerateStructureSystem()
    quad = sys.BakeBLAS(H.Mesh.quad(), tree)
    placements = [((-1.6, 0.0, 0.0), 2.2), ((1.1, -0.2, 0.3), 1.5), ((0.2, -0.55, -1.4), 0.9)]
    for m, (pos, size) in zip(meshes, placements):
        ext = (m.positions.max(axis=0) - m.positions.min(axis=0)).max()
        ctr = (m.positions.max(axis=0) + m.positions.min(axis=0)) * 0.5
        s = size / float(ext)
        b = sys.BakeBLAS(m, tree)
        t = np.asarray(pos, dtype=np.float64) - ctr * s
        sys.AddEntity(b, H.trs(tuple(t), H.quat_axis_angle((0, 1, 0), 25.0), (s, s, s)), albedo=(1, 1, 1), specular=(0.2, 0.2, 0.2))
    # walls: the three emissive quads of the reference sub-scene
    # (Assets/Scenes/SubScenes/RayTracingSubScene.unity:305-307,416-418,567-569), scaled to this scene
    sys.AddEntity(quad, H.trs((4.0, 1.0, 0.0), H.quat_axis_angle((0, 1, 0), 90), (10, 10, 1)), albedo=(1, 1, 1),
                  specular=(0.2, 0.2, 0.2), emission=(0, 0.034409046, 1))
    sys.AddEntity(quad, H.trs((0.0, 1.0, 4.0), H.quat_axis_angle((0, 1, 0), 0), (10, 10, 1)), albedo=(1, 1, 1),
                  specular=(0.2, 0.2, 0.2), emission=(1, 1, 1))
    sys.AddEntity(quad, H.trs((-4.0, 1.0, 0.0), H.quat_axis_angle((0, 1, 0), -90), (10, 10, 1)), albedo=(1, 1, 1),
                  specular=(0.2, 0.2, 0.2), emission=(1, 0, 0))
    sys.AddEntity(quad, H.trs((0, -1.2, 0), _rot_x(90), (10, 10, 1)), albedo=(0.7, 0.7, 0.7), specular=(0.1, 0.1, 0.1))
    buf = sys.OnUpdate()
    sys.close()
    frame = look_at_camera((0.3, 0.6, -5.0), (0.0, -0.1, 0.0), 45.0, width, height)
    return buf, frame, procedural_sky()


def reference_scene(models_dir=None, width=1920, height=1080):
    """The reference's own sample: `peter` + three emissive walls with the sample camera
    (Assets/Scenes/SubScenes/RayTracingSubScene.unity:151,219-221,305-307,416-418,567-569;
    Assets/Scenes/SampleScene.unity:638-640,666-667)."""
    sys = H.BuildAccelerateStructureSystem()
    peter = sys.BakeBLAS(load_model("peter", models_dir))
    quad = sys.BakeBLAS(H.Mesh.quad())
    parent = np.array([20.050472, -2.750208, 25.932217])
    face = dict(albedo=(1, 1, 1), specular=(0.2, 0.2, 0.2))
    sys.AddEntity(peter, H.trs(tuple(parent + (0, 50, 26.7)), s=(50, 50, 50)), **face)
    sys.AddEntity(quad, H.trs(tuple(parent + (53, 50, 0.4)), (0, 0.7071068, 0, 0.7071068), (100, 100, 1)),
                  emission=(0, 0.034409046, 1), **face)
    sys.AddEntity(quad, H.trs(tuple(parent + (3, 50, -40)), (0, 1, 0, 0), (100, 100, 1)), emission=(1, 1, 1), **face)
    sys.AddEntity(quad, H.trs(tuple(parent + (-51.8, 50, -9.5)), (0, -0.7033948, 0, 0.71079946), (100, 100, 1)),
                  emission=(1, 0, 0), **face)
    buf = sys.OnUpdate()
    sys.close()
    frame = H.camera((39.34177, 121.19705, 167.90202), (-0.0028800666, 0.98602176, -0.16570693, -0.017148614),
                     58.2, width / height, 0.3, 1000.0)
    return buf, frame, procedural_sky()


def instanced_scene(width=1920, height=1080, instances=1250, tris_per_blas=40_000, seed=5, tlas_tree=0, glass_every=0):
    """Config 5 shape: many instances of a few ~40 k-triangle meshes with seeded random
    transforms (instances * tris_per_blas instanced triangles).  tlas_tree=1 builds the TLAS
    with the SAH builder (the reference's own TLAS builder gives up beyond about 1 300 objects)."""
    rs = np.random.RandomState(seed)
    sys = H.BuildAccelerateStructureSystem(tlas_tree=tlas_tree)
    rings = max(4, int(round(np.sqrt(tris_per_blas / 4.0))))
    blas = [sys.BakeBLAS(_bumpy_sphere(2 * rings, rings, 100 + k)) for k in range(3)]
    quad = sys.BakeBLAS(H.Mesh.quad())
    sys.AddEntity(quad, H.trs((0, 0, 0), _rot_x(90), (400, 400, 1)), albedo=(0.8, 0.8, 0.8), specular=(0.04, 0.04, 0.04))
    side = int(np.ceil(np.sqrt(instances)))
    for i in range(instances):
        ix, iz = i % side, i // side
        s = float(rs.uniform(0.6, 1.4))
        x = (ix - side / 2) * 2.0 + float(rs.uniform(-0.5, 0.5))
        z = (iz - side / 2) * 2.0 + float(rs.uniform(-0.5, 0.5))
        q = H.quat_axis_angle(rs.uniform(-1, 1, size=3) + 1e-3, float(rs.uniform(0, 360)))
        col = rs.uniform(0.2, 0.9, size=3)
        if glass_every and i % glass_every == 0:      # the "diffuse + dielectric" variant: every n-th instance is tinted glass
            sys.AddEntity(blas[i % 3], H.trs((x, 0.8 * s, z), q, (s, s, s)), albedo=(*(0.5 + 0.5 * col), 1.5), specular=(0, 0, 0, T.MATERIAL_DIELECTRIC))
            continue
        sys.AddEntity(blas[i % 3], H.trs((x, 0.8 * s, z), q, (s, s, s)), albedo=tuple(col), specular=(0.04, 0.04, 0.04))
    buf = sys.OnUpdate()
    sys.close()
    frame = look_at_camera((0.0, 14.0, -side * 1.4), (0.0, 0.0, 0.0), 40.0, width, height)
    return buf, frame, procedural_sky()


def scattered_scene(width=1920, height=1080, instances=5000, tris_per_blas=2000, seed=9):
    """Instances of three meshes scattered through a cube, sizes spread over a factor of 30, with a
    few emissive ones: a top-level tree with no preferred plane (instanced_scene is flat)."""
    rs = np.random.RandomState(seed)
    sys = H.BuildAccelerateStructureSystem(tlas_tree=1)
    rings = max(4, int(round(np.sqrt(tris_per_blas / 4.0))))
    blas = [sys.BakeBLAS(_bumpy_sphere(2 * rings, rings, 200 + k)) for k in range(3)]
    half = 0.9 * float(instances) ** (1.0 / 3.0)
    for i in range(instances):
        s = float(0.1 * 30.0 ** rs.uniform(0.0, 1.0) * 0.3)
        pos = rs.uniform(-half, half, size=3)
        q = H.quat_axis_angle(rs.uniform(-1, 1, size=3) + 1e-3, float(rs.uniform(0, 360)))
        col = rs.uniform(0.2, 0.9, size=3)
        em = (3.0, 3.0, 3.0) if i % 50 == 0 else (0.0, 0.0, 0.0)
        sys.AddEntity(blas[i % 3], H.trs(tuple(pos), q, (s, s * float(rs.uniform(0.5, 1.5)), s)), albedo=tuple(col),
                      specular=(0.04, 0.04, 0.04), emission=em)
    buf = sys.OnUpdate()
    sys.close()
    frame = look_at_camera((0.3 * half, 0.2 * half, -2.6 * half), (0.0, 0.0, 0.0), 40.0, width, height)
    return buf, frame, procedural_sky()


def random_soup(width=640, height=360, seed=0, meshes=4, tris_per_mesh=600, instances=40):
    """Stress scene for parity campaigns: triangle soups (random sizes, slivers, shared vertices, a few
    zero-area triangles) instanced with random rotations, non-uniform and mirrored scales, at
    coordinates from 1e-2 to 1e3; random mix of diffuse / mirror / emissive materials."""
    rs = np.random.RandomState(1000 + seed)
    sys = H.BuildAccelerateStructureSystem(tlas_tree=int(rs.randint(0, 2)))
    blas = []
    for m in range(meshes):
        nt = int(rs.randint(tris_per_mesh // 4, tris_per_mesh + 1))
        if tris_per_mesh <= 8:
            nt = tris_per_mesh                        # (tiny meshes: exactly that many)
        nv = max(3, nt // 2)
        pos = rs.normal(0, 1, size=(nv, 3)).astype(np.float32)
        pos *= np.float32(10.0 ** rs.uniform(-1, 1))
        idx = rs.randint(0, nv, size=(nt, 3)).astype(np.int32)
        near = rs.rand(nt) < 0.6                      # mostly small triangles: neighbours in a shuffled ring
        ring = rs.permutation(nv).astype(np.int32)
        base = rs.randint(0, nv, size=nt)
        idx[near] = np.stack([ring[base[near]], ring[(base[near] + 1) % nv], ring[(base[near] + 2) % nv]], 1)
        idx[: nt // 50, 1] = idx[: nt // 50, 0]       # zero-area triangles
        nrm = pos / np.maximum(np.linalg.norm(pos, axis=1, keepdims=True), 1e-6)
        blas.append(sys.BakeBLAS(H.Mesh(pos, nrm.astype(np.float32), idx), int(rs.randint(0, 2))))
    world = float(10.0 ** rs.uniform(0.5, 3.0))
    for i in range(instances):
        q = H.quat_axis_angle(rs.normal(size=3) + 1e-3, float(rs.uniform(0, 360)))
        s = (10.0 ** rs.uniform(-1.0, 0.5, size=3)) * world * 0.05
        if rs.rand() < 0.3:
            s[int(rs.randint(0, 3))] *= -1.0         # mirrored instance
        t = rs.uniform(-0.5, 0.5, size=3) * world
        kind = rs.rand()
        if kind < 0.6:
            mat = dict(albedo=tuple(rs.uniform(0.2, 0.9, 3)), specular=(0.04, 0.04, 0.04))
        elif kind < 0.85:
            mat = dict(albedo=(0, 0, 0), specular=tuple(rs.uniform(0.5, 1.0, 3)))
        else:
            mat = dict(albedo=(0, 0, 0), specular=(0, 0, 0), emission=tuple(rs.uniform(0.5, 3.0, 3)))
        sys.AddEntity(blas[int(rs.randint(0, meshes))], H.trs(tuple(t), q, tuple(s)), **mat)
    buf = sys.OnUpdate()
    sys.close()
    ang = rs.uniform(0, 2 * np.pi)
    eye = np.array([np.cos(ang), rs.uniform(-0.3, 0.6), np.sin(ang)]) * world * rs.uniform(0.2, 1.4)
    frame = look_at_camera(tuple(eye), tuple(rs.uniform(-0.1, 0.1, 3) * world), float(rs.uniform(30, 80)), width, height)
    return buf, frame, procedural_sky()


def sliver_scene(w, h, seed, coord, inst_scale, n_tris=400, instances=12, distance=8.0):
    """Stress scene for the production kernel's distance culling: slivers (aspect 1e4 .. 3e5) and a few
    wide triangles stored `coord` (1e3 .. 1e5) away from their mesh's origin, under rotated, mirrored,
    non-uniformly scaled instances of overall scale `inst_scale` (1e-3 .. 1e3) that bring them back to
    the world origin; odd seeds look along the horizon (grazing rays), even seeds straight down an
    axis.  At these coordinates binary32 cancellation makes a box's computed entry distance and its
    own triangle's computed hit distance disagree by far more than an ulp."""
    rs = np.random.RandomState(seed)
    sys_ = H.BuildAccelerateStructureSystem(tlas_tree=int(rs.randint(0, 2)))
    blas = []
    for m in range(3):
        base = rs.uniform(-1, 1, size=(n_tris, 3)) * 2.0 + rs.choice([-1, 1], size=3) * coord
        long_dir = rs.normal(size=(n_tris, 3))
        long_dir /= np.linalg.norm(long_dir, axis=1, keepdims=True)
        short_dir = np.cross(long_dir, rs.normal(size=(n_tris, 3)))
        short_dir /= np.linalg.norm(short_dir, axis=1, keepdims=True)
        L = rs.uniform(1.0, 4.0, size=(n_tris, 1))
        aspect = 10.0 ** rs.uniform(4.0, 5.5, size=(n_tris, 1))
        p0, p1, p2 = base, base + long_dir * L, base + long_dir * L * 0.5 + short_dir * (L / aspect)
        pos = np.concatenate([p0, p1, p2]).astype(np.float32)
        idx = np.stack([np.arange(n_tris), np.arange(n_tris) + n_tris, np.arange(n_tris) + 2 * n_tris], 1).astype(np.int32)
        # a few wide triangles so that something is hit behind the slivers
        nrm = np.cross(p1 - p0, p2 - p0)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
        nv = np.concatenate([nrm, nrm, nrm]).astype(np.float32)
        big0 = base[:40]
        bp = np.concatenate([big0, big0 + rs.normal(size=(40, 3)) * 1.5, big0 + rs.normal(size=(40, 3)) * 1.5]).astype(np.float32)
        bi = np.stack([np.arange(40), np.arange(40) + 40, np.arange(40) + 80], 1).astype(np.int32) + len(pos)
        bn = np.tile(np.array([[0, 1, 0]], np.float32), (120, 1))
        mesh = H.Mesh(np.concatenate([pos, bp]), np.concatenate([nv, bn]), np.concatenate([idx, bi]))
        blas.append((sys_.BakeBLAS(mesh, int(rs.randint(0, 2))), base.mean(0)))
    for i in range(instances):
        b, c_local = blas[i % 3]
        q = H.quat_axis_angle(rs.normal(size=3) + 1e-3, float(rs.uniform(0, 360)))
        s = inst_scale * (10.0 ** rs.uniform(-0.3, 0.3, size=3))
        if rs.rand() < 0.3:
            s[int(rs.randint(0, 3))] *= -1
        # place the instance so that its mesh (centred at c_local in local space) lands near the world origin
        m0 = H.trs((0, 0, 0), q, tuple(s)).reshape(4, 4).T
        t = rs.uniform(-1, 1, size=3) * 3.0 * inst_scale - (m0[:3, :3] @ c_local)
        kind = rs.rand()
        mat = dict(albedo=tuple(rs.uniform(0.2, 0.9, 3)), specular=(0.04, 0.04, 0.04)) if kind < 0.6 else (
            dict(albedo=(0, 0, 0), specular=tuple(rs.uniform(0.5, 1.0, 3))) if kind < 0.85 else dict(albedo=(0, 0, 0), specular=(0, 0, 0), emission=(2, 2, 2)))
        sys_.AddEntity(b, H.trs(tuple(t), q, tuple(s)), **mat)
    buf = sys_.OnUpdate()
    sys_.close()
    R = distance * inst_scale
    ang = rs.uniform(0, 2 * np.pi)
    # near-axis and grazing views
    eye = np.array([np.cos(ang) * R, rs.uniform(-0.05, 0.05) * R, np.sin(ang) * R]) if seed % 2 else np.array([R * 1e-4, R, R * 1e-4])
    frame = look_at_camera(tuple(eye), (0.0, 0.0, 0.0), float(rs.uniform(20, 60)), w, h)
    return buf, frame, procedural_sky()


def glass_scene(width=640, height=360):
    """Dielectric extension: two glass spheres (index 1.5 and a tinted 1.33) and a glass cube on a diffuse
    floor next to a diffuse sphere, under an emissive quad.  Glass = specular.w 2, index of refraction
    in albedo.w, tint in albedo.xyz (the reference never reads the fourth components)."""
    sys = H.BuildAccelerateStructureSystem()
    sph = sys.BakeBLAS(H.Mesh.uv_sphere(48, 24, 0.5))
    quad = sys.BakeBLAS(H.Mesh.quad())
    cube = sys.BakeBLAS(H.Mesh.cube())
    sys.AddEntity(quad, H.trs((0, 0, 0), _rot_x(90), (40, 40, 1)), albedo=(0.8, 0.8, 0.8), specular=(0.04, 0.04, 0.04))
    sys.AddEntity(sph, H.trs((-1.2, 0.75, 0), s=(1.5, 1.5, 1.5)), albedo=(1, 1, 1, 1.5), specular=(0, 0, 0, T.MATERIAL_DIELECTRIC))
    sys.AddEntity(sph, H.trs((0.6, 0.5, -0.6), s=(1, 1, 1)), albedo=(0.6, 0.9, 0.7, 1.33), specular=(0, 0, 0, T.MATERIAL_DIELECTRIC))
    sys.AddEntity(cube, H.trs((1.8, 0.5, 0.6), H.quat_axis_angle((0, 1, 0), 30), (1, 1, 1)), albedo=(0.9, 0.9, 1.0, 1.5),
                  specular=(0, 0, 0, T.MATERIAL_DIELECTRIC))
    sys.AddEntity(sph, H.trs((0.2, 0.4, 1.6), s=(0.8, 0.8, 0.8)), albedo=(0.8, 0.3, 0.2), specular=(0.1, 0.1, 0.1))
    sys.AddEntity(quad, H.trs((0, 4, 0.5), _rot_x(-90), (3, 3, 1)), albedo=(0, 0, 0), specular=(0, 0, 0), emission=(3, 3, 3))
    buf = sys.OnUpdate()
    sys.close()
    return buf, look_at_camera((0.3, 1.8, -5.5), (0.2, 0.5, 0), 40.0, width, height), procedural_sky()


def quat_axis_angle_np(axis, degrees):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = np.radians(degrees) / 2
    return np.array([*(a * np.sin(h)), np.cos(h)])


def _quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def needle_comb_scene(w, h, needles=6000, length=4.0, aspect=2.0e4, layers=2, layer_gap=2.0e-3, tilt_deg=25.0, skew_deg=0.002, inplane_deg=41.0, seed=0):
    """Built to strain the production kernels' distance culling beyond its slack: `layers` flat sheets of needle
    triangles (aspect `aspect`: long edges `length`, width length / aspect) lying `layer_gap` apart, the needles
    oblique to the axes within the sheet.  Moeller-Trumbore's determinant on such a triangle is a difference of
    products some `aspect` times larger than itself, so the computed hit distance is off by about 2^-24 times
    that, while the triangle's box is thin along the view and its computed entry distance is good to an ulp:
    the box is entered, as computed, behind the hit.  The sheets lie close enough that such errors decide
    which sheet a ray hits first."""
    rs = np.random.RandomState(50 + seed)
    wd = length / aspect
    pos, idx = [], []
    for layer in range(layers):
        z = layer * layer_gap
        x0 = -0.5 * needles * wd
        base = len(pos)
        for i in range(needles + 1):
            pos.append((x0 + i * wd, -0.5 * length, z))                  # bottom row
        for i in range(needles + 1):
            pos.append((x0 + (i + 0.5) * wd, 0.5 * length, z))           # top row, half a width along
        top = base + needles + 1
        for i in range(needles):
            # (the apex first: Moeller-Trumbore's two edges are then the two long, almost parallel ones)
            idx.append((top + i, base + i + 1, base + i))                # needle pointing up
            idx.append((base + i + 1, top + i, top + i + 1))             # needle pointing down
    # the mesh is stored turned by a very small rotation about an arbitrary axis and shifted: no component of any
    # edge is zero or a round number, yet the needles' boxes stay thin along the view; the instance turns it back
    # ... after a turn within the sheet's own plane, which leaves the boxes as thin but makes the needles' long edges
    # oblique to the axes: the cross products of Moeller-Trumbore then subtract products of nearly equal size
    R = _quat_matrix(quat_axis_angle_np((0.3, -0.8, 0.52), skew_deg)) @ _quat_matrix(quat_axis_angle_np((0, 0, 1), inplane_deg))
    shift = np.array([3.1, -1.7, 2.3])
    pos = (np.asarray(pos, dtype=np.float64) @ R.T + shift).astype(np.float32)
    idx = np.asarray(idx, dtype=np.int32)
    nrm = np.tile((R @ np.array([0.0, 0.0, -1.0])).astype(np.float32), (len(pos), 1))
    sys_ = H.BuildAccelerateStructureSystem()
    comb = sys_.BakeBLAS(H.Mesh(pos, nrm, idx), int(rs.randint(0, 2)))
    quad = sys_.BakeBLAS(H.Mesh.quad())
    span = needles * wd
    # world = Rx(tilt) * R^-1 * (p - shift)
    Rt = _quat_matrix(quat_axis_angle_np((1, 0, 0), tilt_deg)) @ R.T
    m = np.eye(4)
    m[:3, :3] = Rt
    m[:3, 3] = -Rt @ shift
    sys_.AddEntity(comb, m.T.astype(np.float32).reshape(16), albedo=(0.7, 0.7, 0.7), specular=(0.1, 0.1, 0.1))
    sys_.AddEntity(quad, H.trs((0, -2.0, 0), _rot_x(90), (40, 40, 1)), albedo=(0.8, 0.8, 0.8), specular=(0.04, 0.04, 0.04))
    sys_.AddEntity(quad, H.trs((0, 6.0, 0), _rot_x(-90), (6, 6, 1)), albedo=(0, 0, 0), specular=(0, 0, 0), emission=(3, 3, 3))
    buf = sys_.OnUpdate()
    sys_.close()
    # the camera looks along -z at the sheets' front faces from close enough that the comb fills the frame
    dist = 0.6 * span / np.tan(np.radians(20.0))
    frame = look_at_camera((0.0, 0.05 * length, -dist), (0.0, 0.0, 0.0), 40.0, w, h)
    return buf, frame, procedural_sky()
