"""One context over several GPUs (rtamd_ctx_create_multi) through the C ABI.  On a one-GPU box the
device list names GPU 0 twice or three times: every listed entry is a context, a host thread and a
stream of its own, the frame is split and reassembled exactly as on distinct GPUs, and only the
transport differs (RCCL takes one rank per GPU, so peer copies carry the exchange here)."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def scene():
    from raytracing_amd import scenes
    return scenes.random_spheres(333, 190, grid=8)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_tile_sharded_frame_equals_the_one_gpu_frame(gpu_pass, scene, n):
    from raytracing_amd import backend
    buf, frame, sky = scene
    w, h = 333, 190
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    m = backend.RayTracingPass(devices=[0] * n)
    try:
        info = m.MultiInfo()
        assert info["devices"] == n
        assert (info["rccl_ranks"] == 1) if n == 1 else (info["rccl_ranks"] == 0 and "twice" in info["note"])
        m.SetScene(buf)
        m.SetSkyBox(sky)
        for kw in (dict(mode=T.MODE_REF), dict(mode=T.MODE_PT, spp=5, max_bounces=6, seed=9),
                   dict(mode=T.MODE_PT, spp=2, max_bounces=3, seed=4, tile_x=40, tile_y=17, tile_w=200, tile_h=100)):
            one = gpu_pass.Execute(frame, w, h, out_rgba=np.full((h, w, 4), 7.0, np.float32), out_prim_id=np.full((h, w, 2), -9, np.int32),
                                   out_t=np.full((h, w), -3.0, np.float32), **kw)
            many = m.Execute(frame, w, h, out_rgba=np.full((h, w, 4), 7.0, np.float32), out_prim_id=np.full((h, w, 2), -9, np.int32),
                             out_t=np.full((h, w), -3.0, np.float32), **kw)
            assert _same(one["rgba"], many["rgba"]), kw
            assert _same(one["ids"], many["ids"]) and _same(one["t"], many["t"]), kw
            assert one["rays"] == many["rays"] and one["rays"] > 0
        # counters add up over the devices
        c1 = gpu_pass.Execute(frame, w, h, out_rgba=True, counters=True, mode=T.MODE_PT, spp=2, max_bounces=3, seed=1)["counters"]
        cn = m.Execute(frame, w, h, out_rgba=True, counters=True, mode=T.MODE_PT, spp=2, max_bounces=3, seed=1)["counters"]
        assert c1 == cn
        # device-memory output on the first GPU
        import torch
        out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        m.Execute(frame, w, h, out_rgba=out, mode=T.MODE_PT, spp=5, max_bounces=6, seed=9)
        ref = gpu_pass.Execute(frame, w, h, out_rgba=True, mode=T.MODE_PT, spp=5, max_bounces=6, seed=9)
        assert _same(out.cpu().numpy(), ref["rgba"])
    finally:
        m.Dispose()


def test_sample_sharded_frame_and_progressive_accumulation(gpu_pass, scene):
    from raytracing_amd import backend
    buf, frame, sky = scene
    w, h = 333, 190
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    m = backend.RayTracingPass(devices=[0, 0, 0])
    try:
        m.SetScene(buf)
        m.SetSkyBox(sky)
        kw = dict(mode=T.MODE_PT, spp=8, max_bounces=5, seed=21)          # 8 samples over 3 devices: 3 + 3 + 2
        one = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, **kw)
        many = m.Execute(frame, w, h, out_rgba=True, out_prim_id=True, flags=T.FLAG_SHARD_SAMPLES, **kw)
        assert _same(one["ids"], many["ids"]) and one["rays"] == many["rays"]
        # same samples, summed in another order: equal to fp32 rounding of an 8-term sum (stated tolerance)
        scale = np.maximum(np.abs(one["rgba"]), 1e-3)
        assert (np.abs(one["rgba"] - many["rgba"]) / scale).max() < 1e-5
        sums = m.Execute(frame, w, h, out_rgba=True, flags=T.FLAG_SHARD_SAMPLES | T.FLAG_OUTPUT_SUM, **kw)["rgba"]
        assert np.all(sums[..., 3] == 8.0)
        with pytest.raises(backend.BackendError, match="tile sharding"):
            m.Execute(frame, w, h, out_rgba=True, flags=T.FLAG_SHARD_SAMPLES | T.FLAG_ACCUMULATE, **kw)
        with pytest.raises(backend.BackendError, match="shards the frame itself"):
            m.Execute(frame, w, h, out_rgba=True, shard_rank=0, shard_count=2, **kw)
        # running mean over two frames, tiles sharded: every device keeps the sums of its own tiles
        gpu_pass.ResetAccumulation()
        m.ResetAccumulation()
        for k in range(2):
            a = gpu_pass.Execute(frame, w, h, out_rgba=True, mode=T.MODE_PT, spp=2, max_bounces=4, seed=3, sample_offset=2 * k, flags=T.FLAG_ACCUMULATE)
            b = m.Execute(frame, w, h, out_rgba=True, mode=T.MODE_PT, spp=2, max_bounces=4, seed=3, sample_offset=2 * k, flags=T.FLAG_ACCUMULATE)
        assert _same(a["rgba"], b["rgba"])
        # a refresh with moved objects reaches every device
        from raytracing_amd import scenes
        buf2, _, _ = scenes.random_spheres(w, h, grid=8)
        buf2.objects["LocalToWorld"][5][13] += 0.4
        m.SetScene(buf2) if len(buf2.objects) != len(buf.objects) else m.RefreshObjects(buf2)
    finally:
        m.Dispose()


def test_multi_context_rejects_bad_device_lists():
    from raytracing_amd import backend
    with pytest.raises(backend.BackendError):
        backend.RayTracingPass(devices=[])
    with pytest.raises(backend.BackendError, match="device 99"):
        backend.RayTracingPass(devices=[0, 99])


def _distinct_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_distinct_devices() < 2", reason="needs two GPUs: RCCL takes one communicator rank per device")
@pytest.mark.parametrize("sharding", ["tiles", "samples"])
def test_rccl_carries_the_exchange_between_distinct_gpus(gpu_pass, scene, sharding):
    """On a node with several GPUs the gather (ncclSend / ncclRecv) and the reduction (ncclReduce) run inside
    the library: one rank per device, and the frame equals the one-GPU frame (bit for bit with tile sharding,
    to the rounding of the per-pixel sum with sample sharding)."""
    import torch
    from raytracing_amd import backend
    buf, frame, sky = scene
    w, h = 333, 190
    n = min(_distinct_devices(), 8)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    m = backend.RayTracingPass(devices=list(range(n)))
    try:
        info = m.MultiInfo()
        assert info["devices"] == n and info["rccl_ranks"] == n, info
        m.SetScene(buf)
        m.SetSkyBox(sky)
        kw = dict(mode=T.MODE_PT, spp=2 * n, max_bounces=5, seed=33)
        one = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
        many = m.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=T.FLAG_SHARD_SAMPLES if sharding == "samples" else 0, **kw)
        assert m.MultiInfo()["rccl_ranks"] == n, m.MultiInfo()        # (no run-time fall-back happened)
        assert _same(one["ids"], many["ids"]) and _same(one["t"], many["t"]) and one["rays"] == many["rays"]
        if sharding == "tiles":
            assert _same(one["rgba"], many["rgba"])
        else:
            scale = np.maximum(np.abs(one["rgba"]), 1e-3)
            assert (np.abs(one["rgba"] - many["rgba"]) / scale).max() < 1e-5      # fp32 sum of 2n terms in another order
    finally:
        m.Dispose()


def test_exchange_falls_back_to_peer_copies_when_rccl_fails_at_run_time(gpu_pass, scene, monkeypatch):
    """A failing RCCL call must not lose the frame: the communicators are dropped, peer copies carry this and
    every later exchange, and the context says so.  (RTAMD_RCCL_FAIL makes the library treat its first RCCL call
    of the exchange as failed; with one GPU the rehearsal context has no communicator and takes the peer-copy
    path from the start, which is the path the fall-back ends in.)"""
    import torch
    from raytracing_amd import backend
    buf, frame, sky = scene
    w, h = 333, 190
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    n = _distinct_devices()
    devices = list(range(min(n, 4))) if n >= 2 else [0, 0]
    monkeypatch.setenv("RTAMD_RCCL_FAIL", "1")
    m = backend.RayTracingPass(devices=devices)
    try:
        m.SetScene(buf)
        m.SetSkyBox(sky)
        for flags in (0, T.FLAG_SHARD_SAMPLES):
            kw = dict(mode=T.MODE_PT, spp=4, max_bounces=4, seed=2)
            one = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, **kw)
            many = m.Execute(frame, w, h, out_rgba=True, out_prim_id=True, flags=flags, **kw)
            assert _same(one["ids"], many["ids"]) and one["rays"] == many["rays"]
            if flags == 0:
                assert _same(one["rgba"], many["rgba"])
            info = m.MultiInfo()
            assert info["rccl_ranks"] == 0 and "peer copies" in info["note"], info
    finally:
        m.Dispose()
