#!/usr/bin/env python3
"""Headline benchmark: path-traced Mrays/s on one node of MI355X GPUs.

A step is one full render of the workload: every pixel, every sample, every bounce, scene
and framebuffers resident in HBM.  At N=1 the workload is BASELINE.json configs[1]: the
random-spheres scene (floor + 484 instances of one sphere BLAS) at 1920x1080, 64 spp,
8 bounces.  With N GPUs (one process per GPU, launched by torch.distributed.run) each rank
renders the same image with its own 64-sample range and the per-pixel sums are all-reduced
with RCCL inside the timed region (weak scaling: per-GPU work is fixed).

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
FP32_PEAK_TFLOPS = 157.3       # vector FP32, spec (packed FMA)


def build_workload(name, width, height, tree=0):
    from raytracing_amd import scenes
    if name == "spheres":
        return scenes.random_spheres(width, height, tree=tree), "random-spheres: floor + 484 instanced spheres (one 960-triangle BLAS)"
    if name == "mesh":
        return scenes.mesh_scene(width, height, tree=tree), "three procedural ~1M-triangle meshes + 4 quads"
    if name == "instanced":
        return scenes.instanced_scene(width, height), "1 250 instances of three ~40 k-triangle meshes (50 M instanced triangles) on a floor"
    if name == "few":
        return scenes.few_spheres(width, height), "few-sphere plumbing scene"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(buf, frame, sky, width, height, spp, bounces, seed):
    """Time the scalar CPU oracle on a bounded sample of the same workload."""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 16)
    # sample: the same camera, scene and resolution at half the sample count (about 10-20 s on 16 cores)
    sw, sh, sspp = width, height, max(1, spp // 2)
    t0 = time.time()
    r = O.render(buf, frame, sw, sh, sky=sky, mode=1, spp=sspp, max_bounces=bounces, seed=seed, nthreads=cores,
                 want_ids=False, want_t=False)
    dt = time.time() - t0
    c = r["counters"]
    # and one core alone on 2 spp of the same frame (a few seconds)
    t1 = time.time()
    r1 = O.render(buf, frame, sw, sh, sky=sky, mode=1, spp=min(2, sspp), max_bounces=bounces, seed=seed, nthreads=1,
                  want_ids=False, want_t=False)
    dt1 = time.time() - t1
    return {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{sw}x{sh}, {sspp} spp, {bounces} bounces of the same scene and camera ({c['rays']} ray segments, {dt:.1f} s)",
            "single_core_value": r1["counters"]["rays"] / dt1 / 1e6,
            "nodes_per_ray": c["nodes"] / max(1, c["rays"]), "tris_per_ray": c["tris"] / max(1, c["rays"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="spheres")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--tree", type=int, default=0, help="BLAS builder of the synthetic scene: 0 = reference median split, 1 = SAH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true")
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

    (buf, frame, sky), desc = build_workload(args.workload, args.width, args.height, args.tree)
    W, Hh = args.width, args.height
    rt = backend.RayTracingPass(dev_index)
    # kernels, the RCCL all-reduce and the normalisation all run on torch's current stream
    rt.SetStream(torch.cuda.current_stream().cuda_stream)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((Hh, W, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        # each rank renders its own sample range; sums are combined below
        flags = T.FLAG_OUTPUT_SUM if world > 1 else 0
        r = rt.Execute(frame, W, Hh, out_rgba=out, mode=T.MODE_PT, spp=args.spp, max_bounces=args.bounces,
                       seed=args.seed, sample_offset=rank * args.spp, flags=flags)
        if world > 1:
            D.combine_samples(out)     # RCCL all-reduce (SUM) of the sum buffer over xGMI, then / sample count
        return r

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = 0
    kernel_ms = 0.0
    handover_ms = 0.0
    trace_ms = 0.0
    for _ in range(args.steps):
        r = step()
        rays += r["rays"]
        kernel_ms += r["kernel_ms"]
        handover_ms += r["handover_ms"]
        trace_ms += r["trace_ms"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays], dtype=torch.int64, device=dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays = int(rr.item())
    value = rays / elapsed / 1e6
    rays_per_launch_local = r["rays"]
    # the path-tracing kernel alone (what rocprofv3 lists as rt::kPool), the summation of the samples
    # behind it, and the hand-over pass
    handover_ms_avg = handover_ms / max(1, args.steps)
    kernel_ms_avg = trace_ms / max(1, args.steps)
    sum_ms_avg = kernel_ms / max(1, args.steps) - handover_ms_avg - kernel_ms_avg
    kernel_name = "rt::kFast<PT>" if os.environ.get("RTAMD_PIPELINE") == "persistent" or os.environ.get("RTAMD_STATS") else "rt::kPool"

    result = None
    if rank == 0:
        # algorithmic work of ONE launch on this rank, counted in the reference's traversal order
        roofline = None
        if not args.no_counters:
            c = rt.Execute(frame, W, Hh, out_rgba=out, counters=True, mode=T.MODE_PT, spp=args.spp, max_bounces=args.bounces,
                           seed=args.seed, sample_offset=0)["counters"]
            alg_bytes = 116 * c["nodes"] + 48 * c["tris"] + 180 * c["instances"] + 36 * c["hits"]
            alg_flops = 100 * c["nodes"] + 56 * c["tris"] + 30 * c["instances"]
            achieved = alg_bytes / (kernel_ms_avg * 1e-3) / 1e9
            # counters of the same launch, collected offline with rocprofv3 --pmc (tools/gpu_pmc.sh)
            traffic, pmc_extra = None, None
            pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    if rec.get("workload") == args.workload and rec.get("spp") == args.spp and rec.get("width") == W:
                        traffic = rec.get("hbm_bytes_per_launch")
                        pmc_extra = {k: rec[k] for k in ("simd_valu_issue_frac", "valu_lane_utilisation", "l2_hit_rate",
                                                         "l1_accesses_per_cu_cycle", "wave_cycles_waiting_frac") if k in rec}
                except Exception:
                    traffic, pmc_extra = None, None
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "kernel": kernel_name, "kernel_ms": kernel_ms_avg, "sample_sum_ms": sum_ms_avg,
                        "handover_ms": handover_ms_avg, "handover_pixels": r["handover_pixels"],
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "bytes_per_ray": alg_bytes / max(1, c["rays"]),
                        "nodes_per_ray": c["nodes"] / max(1, c["rays"]), "tris_per_ray": c["tris"] / max(1, c["rays"]),
                        "instances_per_ray": c["instances"] / max(1, c["rays"]),
                        "valu_tflops_algorithmic": alg_flops / (kernel_ms_avg * 1e-3) / 1e12,
                        "valu_frac_of_fp32_peak": alg_flops / (kernel_ms_avg * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "kernel_mrays_per_s": rays_per_launch_local / (kernel_ms_avg * 1e-3) / 1e6,
                        "pmc_offline": pmc_extra}
        base = None
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(buf, frame, sky, W, Hh, args.spp, args.bounces, args.seed)
        result = {
            "metric": "Mrays/s (path-traced, incl. bounces)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}, {W}x{Hh}, {args.spp} spp/GPU, {args.bounces} bounces",
                       "resolution": [W, Hh], "spp_per_gpu": args.spp, "max_bounces": args.bounces,
                       "parallelism": f"sample-sharded x{world}, RCCL all-reduce of the sum buffer" if world > 1 else "single GPU",
                       "samples_per_s": W * Hh * args.spp * world / (elapsed / max(1, args.steps))},
            "roofline": roofline, "cpu_baseline": base,
        }
        print(json.dumps(result), flush=True)
    rt.Dispose()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
