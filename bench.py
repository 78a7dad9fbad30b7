#!/usr/bin/env python3
"""Headline benchmark: path-traced Mrays/s on one node of MI355X GPUs.

A step is one full render of the workload: every pixel, every sample, every bounce, scene
and framebuffers resident in HBM.

  N = 1   BASELINE.json configs[2]: the triangle-mesh scene built from the reference's
          Assets/Models (peter / bunny / cow subdivided to 1 051 936 triangles, each mesh its
          own BLAS, plus the sub-scene's emissive walls) at 1920x1080, 256 spp, 8 bounces.
          Measured as well and reported inside the same line ("other_workloads"): the random-spheres
          scene of configs[1] (1080p, 64 spp), the 50 M-instanced-triangle scene of configs[4] (1080p, 64 spp),
          and REF mode -- what the reference's shader renders, one primary and one mirror ray per pixel
          (Assets/Shaders/RayTracingShader.compute:413-429) -- as milliseconds per rtamd_render call.
  N > 1   BASELINE.json configs[3]: the same scene at 3840x2160, 1024 spp in total, STRONG
          scaling by tile sharding (one process per GPU, launched by torch.distributed.run):
          rank r renders the 32x32-pixel tiles t with t % N == r into a zeroed frame, and the
          frames are all-reduced (SUM) with RCCL over xGMI inside the timed region; the
          result is bit-identical to the one-GPU image.  (--sharding samples selects sample
          sharding instead: every rank renders total_spp / N samples of every pixel.)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec
FP32_PEAK_TFLOPS = 157.3       # vector FP32, spec: 256 CUs x 128 lanes x 2 (FMA) x 2.4 GHz
VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 4.0   # wave instructions per ns: 1024 SIMDs, one wave64 instruction per 4 cycles at 2.4 GHz

DEFAULT_SPP = {"mesh": 256, "spheres": 64, "instanced": 64, "few": 4}


def build_workload(name, width, height, tree=0):
    from raytracing_amd import scenes
    if name == "spheres":
        return scenes.random_spheres(width, height, tree=tree), "random-spheres: floor + 484 instanced spheres (one 960-triangle BLAS)"
    if name == "mesh":
        s = scenes.mesh_scene(width, height, tree=tree)
        what = ("Assets/Models peter+bunny+cow subdivided 2/3/2x" if scenes.have_models() else "three procedural meshes")
        return s, f"triangle-mesh scene: {what} = {len(s[0].triangles)} triangles in 3 BLAS + 4 quads"
    if name == "instanced":
        return scenes.instanced_scene(width, height), "1 250 instances of three ~40 k-triangle meshes (50 M instanced triangles) on a floor"
    if name == "few":
        return scenes.few_spheres(width, height), "few-sphere plumbing scene"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(buf, frame, sky, width, height, spp, bounces, seed):
    """Time the scalar CPU oracle on a bounded sample of the same workload (about 15 s on 16 cores)."""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 16)
    sspp = max(1, min(32, spp // 8))
    t0 = time.time()
    r = O.render(buf, frame, width, height, sky=sky, mode=1, spp=sspp, max_bounces=bounces, seed=seed, nthreads=cores,
                 want_ids=False, want_t=False)
    dt = time.time() - t0
    c = r["counters"]
    # and one core alone on a quarter-size tile of one sample (a few seconds)
    t1 = time.time()
    r1 = O.render(buf, frame, width, height, sky=sky, mode=1, spp=1, max_bounces=bounces, seed=seed, nthreads=1,
                  tile=(width // 4, height // 4, width // 2, height // 2), want_ids=False, want_t=False)
    dt1 = time.time() - t1
    return {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{width}x{height}, {sspp} spp, {bounces} bounces of the same scene and camera ({c['rays']} ray segments, {dt:.1f} s)",
            "single_core_value": r1["counters"]["rays"] / dt1 / 1e6,
            "nodes_per_ray": c["nodes"] / max(1, c["rays"]), "tris_per_ray": c["tris"] / max(1, c["rays"])}


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except Exception:
        return None


def roofline_for(rt, frame, W, Hh, out, args, workload, spp, kernel_ms_avg, sum_ms_avg, handover_ms_avg, last, kernel_name):
    """Roofline object of one workload: algorithmic work (counted in the reference's traversal
    order by the reference-order kernel) over the live duration of the dominant kernel, against
    the ceiling that binds it, plus the other ceilings."""
    from raytracing_amd import types as T
    # counters depend on the sample count only through the number of paths: count 8 spp and scale
    cspp = min(spp, 8)
    c = rt.Execute(frame, W, Hh, out_rgba=out, counters=True, mode=T.MODE_PT, spp=cspp, max_bounces=args.bounces,
                   seed=args.seed, sample_offset=0)["counters"]
    per_ray = {k: c[k] / max(1, c["rays"]) for k in ("nodes", "tris", "instances", "hits")}
    rays = last["rays"]                              # ray segments of one timed step on this rank
    alg_bytes = rays * (116 * per_ray["nodes"] + 48 * per_ray["tris"] + 180 * per_ray["instances"] + 36 * per_ray["hits"])
    alg_flops = rays * (100 * per_ray["nodes"] + 56 * per_ray["tris"] + 30 * per_ray["instances"])
    secs = kernel_ms_avg * 1e-3
    tflops = alg_flops / secs / 1e12
    # offline evidence of the same step: rocprofv3 --pmc passes (tools/gpu_pmc.sh, tools/summarise_pmc.py)
    # and the streaming ceilings of the cache levels (tools/l2_stream.hip)
    # (kept only when the record was measured on the kernels that are running: the library and the record both
    # carry the fingerprint of the device sources)
    from raytracing_amd import backend
    pmc = None
    rec = load_json("pmc_step.json")
    lib_sha = backend.source_sha()
    pmc_sha = (rec or {}).get("kernel_source_sha")
    pmc_current = bool(rec) and pmc_sha == lib_sha
    if rec and pmc_current:
        for r in rec.get("workloads", []):
            if r.get("workload") == workload and r.get("spp") == spp and r.get("width") == W and r.get("height") == Hh:
                pmc = r
    ceil = load_json("r02_stream_ceilings.json")
    ceilings = {"fp32_valu": {"achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS,
                              "what": "algorithmic flops (100/node + 56/triangle + 30/instance, reference-order counts) / kernel time"}}
    traffic = None
    if pmc:
        traffic = pmc.get("hbm_bytes_per_step")
        if traffic:
            gbs = traffic / secs / 1e9
            ceilings["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "what": "fabric bytes of the step (PMC FETCH_SIZE x2 + WRITE_SIZE) / kernel time"}
        if pmc.get("valu_wave_insts_per_step"):
            gi = pmc["valu_wave_insts_per_step"] / secs / 1e9
            ceilings["valu_issue"] = {"achieved": gi, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
                                      "frac": gi / VALU_ISSUE_PEAK_GINST, "lane_utilisation": pmc.get("valu_lane_utilisation"),
                                      "what": "VALU wave instructions of the step (PMC SQ_INSTS_VALU) / kernel time against 1024 SIMDs x 1 per 4 cycles"}
        if pmc.get("l2_read_bytes_per_step") and ceil:
            # gather ceiling of the cache level the scene's nodes and triangles live in
            level = {"spheres": "l2_0.2MB", "instanced": "mall_84MB"}.get(workload, "mall_84MB")
            # one node of the micro-benchmark = one 128-byte line request (it reads 112 bytes of it)
            peak = ceil.get(level, {}).get("gather_nodes_per_s", 0.0) * 128.0 / 1e9
            if peak:
                gbs = pmc["l2_read_bytes_per_step"] / secs / 1e9
                ceilings["l2_gather"] = {"achieved": gbs, "peak": peak, "unit": "GB/s", "frac": gbs / peak,
                                         "write_requests_per_s": pmc.get("l2_write_requests_per_step", 0.0) / secs,
                                         "what": f"L1->L2 read bytes of the step (PMC TCP_TCC_READ_REQ x 128 B) / kernel time against the measured rate of "
                                                 f"per-lane 128-byte line gathers from a {level.split('_')[1]} working set (tools/l2_stream.hip: lines per second x 128 B)"}
    # the ceiling that binds is the one with the largest fraction; FP32 VALU is what the contract prices
    binding = max(ceilings.items(), key=lambda kv: kv[1]["frac"])[0]
    v = ceilings["fp32_valu"]
    return {"bound": "valu", "achieved": v["achieved"], "peak": v["peak"], "unit": v["unit"], "frac": v["frac"], "traffic": traffic,
            "kernel": kernel_name, "kernel_ms": kernel_ms_avg, "sample_sum_ms": sum_ms_avg,
            "handover_ms": handover_ms_avg, "handover_pixels": last["handover_pixels"],
            "binding_ceiling": binding, "ceilings": ceilings,
            "algorithmic_flops_per_step": alg_flops, "algorithmic_bytes_per_step": alg_bytes,
            "throughput_equivalent_GBps": alg_bytes / secs / 1e9,
            "bytes_per_ray": alg_bytes / max(1, rays), "flops_per_ray": alg_flops / max(1, rays),
            "nodes_per_ray": per_ray["nodes"], "tris_per_ray": per_ray["tris"], "instances_per_ray": per_ray["instances"],
            "counted_on": f"{cspp} spp of the same frame, scaled by ray segments",
            "pmc_offline": ({k: pmc[k] for k in ("simd_valu_issue_frac", "valu_lane_utilisation", "l2_hit_rate", "l1_accesses_per_cu_cycle",
                                                 "wave_cycles_waiting_frac", "salu_per_valu", "source") if k in pmc} if pmc else None),
            "pmc_record": {"kernel_source_sha": pmc_sha, "library_source_sha": lib_sha, "current": pmc_current,
                           "note": None if pmc_current else "profiles/pmc_step.json was measured on other kernel sources than the library that ran: "
                                                            "its counter-derived ceilings and `traffic` are left out of this line"}}


def timed_steps(step, steps, warmup, sync, barrier):
    for _ in range(warmup):
        step()
    barrier()
    sync()
    t0 = time.perf_counter()
    acc = {"rays": 0, "kernel_ms": 0.0, "handover_ms": 0.0, "trace_ms": 0.0}
    r = None
    for _ in range(steps):
        r = step()
        for k in acc:
            acc[k] += r[k]
    sync()
    barrier()
    return time.perf_counter() - t0, acc, r


def ref_mode_frames(rt, frame, W, Hh, out, frames=100):
    """REF mode: what the reference's shader renders (one primary and one mirror ray per covered pixel,
    Assets/Shaders/RayTracingShader.compute:413-429), one blocking rtamd_render call per frame as
    CustomRenderPass.Execute makes it (RayTracingPass.cs:101), frame in device memory."""
    import torch
    from raytracing_amd import types as T
    for _ in range(5):
        rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_REF)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km = hm = 0.0
    r = None
    each = []
    for _ in range(frames):
        t1 = time.perf_counter()
        r = rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_REF)
        each.append((time.perf_counter() - t1) * 1e3)
        km += r["kernel_ms"]
        hm += r["handover_ms"]
    torch.cuda.synchronize()
    mean_ms = (time.perf_counter() - t0) / frames * 1e3
    each.sort()
    # (the median of the calls: on a fresh box one call in a few hundred takes tens of milliseconds on the host side,
    # which a mean over 100 frames carries as +0.4 ms; mean and slowest call are reported beside it)
    ms = each[len(each) // 2]
    return {"resolution": [W, Hh], "frames": frames, "ms_per_render_call": ms, "ms_mean_call": mean_ms, "ms_slowest_call": each[-1],
            "kernels_ms": km / frames,
            "handover_ms": hm / frames, "handover_pixels": r["handover_pixels"], "ray_segments": r["rays"],
            "value": r["rays"] / ms / 1e3, "unit": "Mrays/s"}


def frames_in_flight(rt, frame, W, Hh, out, spp, args, dev, steps):
    """The same step through rtamd_render_begin / rtamd_render_wait, two frames in flight (a host that renders frame
    after frame, RayTracingPass.cs:62-109): the next frame's kernel runs while the last paths of this one are
    handed over and summed.  Reported beside the blocking figure, which stays the line's value."""
    import torch
    from raytracing_amd import types as T
    outs = [out, torch.zeros_like(out)]
    kw = dict(mode=T.MODE_PT, spp=spp, max_bounces=args.bounces, seed=args.seed)
    try:
        rt.ExecuteBegin(frame, W, Hh, out_rgba=outs[0], **kw)
        rt.ExecuteWait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = 0
        rt.ExecuteBegin(frame, W, Hh, out_rgba=outs[0], **kw)
        for k in range(1, steps):
            rt.ExecuteBegin(frame, W, Hh, out_rgba=outs[k & 1], **kw)
            rays += rt.ExecuteWait()["rays"]
        rays += rt.ExecuteWait()["rays"]
        torch.cuda.synchronize()
        e = time.perf_counter() - t0
    except Exception as ex:      # (a frame too large for one batch of paths: the blocking call is the only way)
        return {"unavailable": str(ex)}
    return {"value": rays / e / 1e6, "unit": "Mrays/s", "ms_per_step": e / steps * 1e3, "steps": steps,
            "what": "rtamd_render_begin / rtamd_render_wait, two frames in flight, same frames as the blocking steps"}


def pt_workload(rt, name, args, dev, kernel_name, steps):
    """One more path-traced workload on the same context: scene up, timed steps, roofline."""
    import torch
    from raytracing_amd import types as T
    (buf, frame, sky), desc = build_workload(name, 1920, 1080, args.tree)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev)
    spp = DEFAULT_SPP[name]

    def step():
        return rt.Execute(frame, 1920, 1080, out_rgba=out, mode=T.MODE_PT, spp=spp, max_bounces=args.bounces, seed=args.seed)
    e, a, last = timed_steps(step, steps, 1, torch.cuda.synchronize, lambda: None)
    rec = {"workload": f"{desc}, 1920x1080, {spp} spp, {args.bounces} bounces", "value": a["rays"] / e / 1e6,
           "unit": "Mrays/s", "ms_per_step": e / steps * 1e3, "steps": steps,
           "two_frames_in_flight": frames_in_flight(rt, frame, 1920, 1080, out, spp, args, dev, max(steps, 3)),
           "roofline": None if args.no_counters else roofline_for(
               rt, frame, 1920, 1080, out, args, name, spp, a["trace_ms"] / steps,
               a["kernel_ms"] / steps - a["handover_ms"] / steps - a["trace_ms"] / steps, a["handover_ms"] / steps, last, kernel_name)}
    return rec, frame, out


def secondary_workloads(rt, frame, out, args, dev, kernel_name, roofline):
    """The other configurations of the one-GPU line, on the same context (the mesh scene is still up on entry)."""
    import torch
    from raytracing_amd import types as T
    steps = max(2, args.steps)
    others = {}
    others["mesh_two_frames_in_flight"] = frames_in_flight(rt, frame, 1920, 1080, out, args.spp or DEFAULT_SPP["mesh"], args, dev, max(steps, 3))
    ref = {"what": "REF mode = the reference shader's own image (primary ray + one mirror ray per covered pixel, RayTracingShader.compute:413-429): "
                   "wall-clock milliseconds per blocking rtamd_render call (median of the timed calls; ms_mean_call and ms_slowest_call beside it), frame in device memory; handover_ms is the part spent in the "
                   "reference-order kernel on the pixels the production kernel passed on"}
    ref["mesh_1080p"] = ref_mode_frames(rt, frame, 1920, 1080, out)
    out4 = torch.zeros((2160, 3840, 4), dtype=torch.float32, device=dev)      # (same aspect: the same frame record)
    ref["mesh_4k"] = ref_mode_frames(rt, frame, 3840, 2160, out4, frames=50)
    del out4
    if roofline is not None and args.tree == 0:
        # BASELINE configs[2] says "with SAH BVH": the reference-order work counts of the same scene when the
        # caller's trees are SAH trees instead of the reference's median-split ones (kernel time does not depend on
        # the caller's trees, the library rebuilds them; the algorithmic counts do)
        (bs, fs, ss), _ = build_workload("mesh", 1920, 1080, 1)
        rt.SetScene(bs)
        c = rt.Execute(fs, 1920, 1080, out_rgba=out, counters=True, mode=T.MODE_PT, spp=8, max_bounces=args.bounces, seed=args.seed)["counters"]
        n = max(1, c["rays"])
        per = {k: c[k] / n for k in ("nodes", "tris", "instances", "hits")}
        fl = 100 * per["nodes"] + 56 * per["tris"] + 30 * per["instances"]
        by = 116 * per["nodes"] + 48 * per["tris"] + 180 * per["instances"] + 36 * per["hits"]
        roofline["caller_sah_trees"] = {"nodes_per_ray": per["nodes"], "tris_per_ray": per["tris"], "instances_per_ray": per["instances"],
                                        "flops_per_ray": fl, "bytes_per_ray": by,
                                        "fp32_valu_frac_on_these_counts": fl * (roofline["algorithmic_flops_per_step"] / max(1e-9, roofline["flops_per_ray"]))
                                        / (roofline["kernel_ms"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                        "what": "reference-order counts of the same frame (8 spp) with the caller's BLAS and TLAS built by the SAH "
                                                "builder (--tree 1) instead of the reference's median split"}
        del bs
    others["spheres"], frame2, out2 = pt_workload(rt, "spheres", args, dev, kernel_name, steps)
    ref["spheres_1080p"] = ref_mode_frames(rt, frame2, 1920, 1080, out2)
    others["instanced"], _, _ = pt_workload(rt, "instanced", args, dev, kernel_name, steps)
    others["ref"] = ref
    return others


def frame_equals_one_gpu(rt, frame, W, Hh, out, args, dev, rank, world, multi, inproc, buf, sky):
    """Before the timed region of a run over several GPUs: the same frame at 4 samples per pixel rendered sharded
    (exactly as the timed steps shard it) and whole on one GPU; with tile sharding the two must be the same bits.
    Returns the fields the JSON line reports (on rank 0; None elsewhere)."""
    import hashlib
    import torch
    import torch.distributed as dist
    from raytracing_amd import backend, distributed as D, types as T
    kw = dict(mode=T.MODE_PT, spp=4 * max(1, world if multi else args.gpus), max_bounces=args.bounces, seed=args.seed)
    many = torch.zeros_like(out)
    if inproc:
        rt.Execute(frame, W, Hh, out_rgba=many, flags=T.FLAG_SHARD_SAMPLES if args.sharding == "samples" else 0, **kw)
        mi = rt.MultiInfo()
        extra = {"rccl_ranks": mi["rccl_ranks"], "exchange_ms": mi["exchange_ms"]}
    elif args.sharding == "tiles":
        rt.Execute(frame, W, Hh, out_rgba=many, **D.tile_shard_kwargs(rank, world), **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D.combine_tiles([many])
        torch.cuda.synchronize()
        extra = {"rccl_ranks": dist.get_world_size(), "exchange_ms": (time.perf_counter() - t0) * 1e3}
    else:
        per = kw["spp"] // world
        k2 = dict(kw, spp=per, sample_offset=rank * per, flags=T.FLAG_OUTPUT_SUM)
        rt.Execute(frame, W, Hh, out_rgba=many, **k2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D.combine_samples(many)
        torch.cuda.synchronize()
        extra = {"rccl_ranks": dist.get_world_size(), "exchange_ms": (time.perf_counter() - t0) * 1e3}
    if rank != 0:
        return None
    if inproc:
        single = backend.RayTracingPass(0)
        single.SetScene(buf)
        single.SetSkyBox(sky)
    else:
        single = rt
    one = torch.zeros_like(out)
    single.Execute(frame, W, Hh, out_rgba=one, **kw)
    torch.cuda.synchronize()
    if inproc:
        single.Dispose()
    a, b = many.cpu().numpy(), one.cpu().numpy()
    same_bits = hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
    rel = float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())
    extra.update({"image_equal_to_one_gpu": bool(same_bits) if args.sharding == "tiles" else bool(rel < 1e-5),
                  "image_check": (f"{kw['spp']} spp of the timed frame, {args.sharding}-sharded against one GPU: "
                                  + ("same bits" if same_bits else f"largest relative difference {rel:.3g}"))})
    return extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mesh")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel in total (default: 256 on one GPU, 1024 on several)")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--tree", type=int, default=0, help="BLAS builder of the synthetic scene: 0 = reference median split, 1 = SAH")
    ap.add_argument("--sharding", default="tiles", choices=["tiles", "samples"])
    ap.add_argument("--single-process", action="store_true",
                    help="with --gpus N and no launcher: one process, one context over GPUs 0..N-1 (rtamd_ctx_create_multi), "
                         "tile-sharded, frame gathered onto GPU 0 with RCCL inside the library")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the random-spheres measurement inside the one-GPU line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import backend, distributed as D, types as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # RTAMD_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend, to rehearse the N>1 code
    # path on a one-GPU box (numbers from such a run mean nothing).
    rehearse = os.environ.get("RTAMD_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    multi = world > 1
    inproc = args.single_process and world == 1 and args.gpus > 1      # several GPUs inside one context
    if inproc:
        multi_shape = True
    else:
        multi_shape = multi
    W = args.width or (3840 if multi_shape and args.workload == "mesh" else 1920)
    Hh = args.height or (2160 if multi_shape and args.workload == "mesh" else 1080)
    spp = args.spp or (1024 if multi_shape and args.workload == "mesh" else DEFAULT_SPP.get(args.workload, 64))
    if multi and args.sharding == "samples" and spp % world:
        raise SystemExit("--spp must be a multiple of the GPU count with --sharding samples")

    (buf, frame, sky), desc = build_workload(args.workload, W, Hh, args.tree)
    if inproc:
        rt = backend.RayTracingPass(devices=[0] * args.gpus if rehearse else list(range(args.gpus)))
    else:
        rt = backend.RayTracingPass(dev_index)
        # kernels, the RCCL all-reduce and the normalisation all run on torch's current stream
        rt.SetStream(torch.cuda.current_stream().cuda_stream)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((Hh, W, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        if inproc:
            return rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_PT, spp=spp, max_bounces=args.bounces, seed=args.seed,
                              flags=T.FLAG_SHARD_SAMPLES if args.sharding == "samples" else 0)
        if not multi:
            return rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_PT, spp=spp, max_bounces=args.bounces, seed=args.seed)
        if args.sharding == "tiles":
            out.zero_()                                         # pixels of other ranks' tiles must be 0 for the SUM
            r = rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_PT, spp=spp, max_bounces=args.bounces, seed=args.seed,
                           **D.tile_shard_kwargs(rank, world))
            D.combine_tiles([out])                              # RCCL all-reduce (SUM) over xGMI: every rank ends with the whole frame
            return r
        per = spp // world
        r = rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_PT, spp=per, max_bounces=args.bounces, seed=args.seed,
                       sample_offset=rank * per, flags=T.FLAG_OUTPUT_SUM)
        D.combine_samples(out)                                  # RCCL all-reduce (SUM) of the sum buffer, then / sample count
        return r

    def barrier():
        if multi:
            dist.barrier()

    check = None
    if multi or inproc:
        check = frame_equals_one_gpu(rt, frame, W, Hh, out, args, dev, rank, world, multi, inproc, buf, sky)
    elapsed, acc, last = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, barrier)
    rays = acc["rays"]
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays], dtype=torch.int64, device=dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays = int(rr.item())
    value = rays / elapsed / 1e6
    n = max(1, args.steps)
    # the path-tracing kernel alone (what rocprofv3 lists as rt::kPool: all its launches of one step), the
    # summation of the samples behind it, and the hand-over pass
    handover_ms_avg = acc["handover_ms"] / n
    kernel_ms_avg = acc["trace_ms"] / n
    sum_ms_avg = acc["kernel_ms"] / n - handover_ms_avg - kernel_ms_avg
    kernel_name = "rt::kFast<PT>" if os.environ.get("RTAMD_PIPELINE") == "persistent" else "rt::kPool"

    result = None
    if rank == 0:
        roofline = None
        if not args.no_counters:
            roofline = roofline_for(rt, frame, W, Hh, out, args, args.workload, spp, kernel_ms_avg, sum_ms_avg, handover_ms_avg, last, kernel_name)
        others = None
        if not multi and not inproc and not args.no_secondary and args.workload == "mesh":
            others = secondary_workloads(rt, frame, out, args, dev, kernel_name, roofline)
        base = None
        if not args.no_cpu_baseline and not multi:
            base = cpu_baseline(buf, frame, sky, W, Hh, spp, args.bounces, args.seed)
        if inproc:
            mi = rt.MultiInfo()
            par = (f"one process, one context over {mi['devices']} GPUs, {args.sharding}-sharded, exchange inside the library: "
                   + (f"RCCL, {mi['rccl_ranks']} ranks" if mi["rccl_ranks"] else f"peer copies ({mi['note']})") + f", {mi['exchange_ms']:.3f} ms per frame")
        elif multi:
            par = (f"tile-sharded x{world} (32x32 tiles, t % {world}), RCCL all-reduce of the frame, {dist.get_world_size()} ranks" if args.sharding == "tiles"
                   else f"sample-sharded x{world}, RCCL all-reduce of the sum buffer, {dist.get_world_size()} ranks")
        else:
            par = "single GPU"
        result = {
            "metric": "Mrays/s (path-traced, incl. bounces)", "value": value, "unit": "Mrays/s",
            "n_gpus": args.gpus if inproc else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / n * 1e3, "higher_is_better": True, "scaling": "strong" if (multi or inproc) else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}, {W}x{Hh}, {spp} spp in total, {args.bounces} bounces",
                       "resolution": [W, Hh], "spp_total": spp, "max_bounces": args.bounces, "parallelism": par,
                       "ray_segments_per_step": rays // n,
                       "samples_per_s": W * Hh * spp / (elapsed / n)},
            "roofline": roofline, "cpu_baseline": base,
        }
        if check is not None:
            result.update(check)
        if inproc:
            mi = rt.MultiInfo()
            result.update({"rccl_ranks": mi["rccl_ranks"], "exchange_ms": mi["exchange_ms"], "exchange_note": mi["note"] or None})
        if others:
            result["other_workloads"] = others
        print(json.dumps(result), flush=True)
    rt.Dispose()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
