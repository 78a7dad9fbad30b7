"""The multi-GPU exchange step on CPU: two processes, gloo backend.  The renderer behind
`execute` is the CPU oracle here (the GPU library needs a GPU); what is under test is the
sharding parameters and the all-reduce composition in raytracing_amd.distributed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H_ = 96, 54


def _oracle_execute(buf, sky):
    from oracle import oracle as O

    def execute(frame, width, height, out_rgba=None, out_prim_id=None, **ext):
        rgba = out_rgba.numpy() if out_rgba is not None else None
        ids = out_prim_id.numpy() if out_prim_id is not None else None
        shard = (ext.get("shard_rank", 0), ext.get("shard_count", 0))
        O.render(buf, frame, width, height, sky=sky, mode=ext.get("mode", 0), spp=ext.get("spp", 1),
                 max_bounces=ext.get("max_bounces", 0), seed=ext.get("seed", 0), sample_offset=ext.get("sample_offset", 0),
                 shard=shard if shard[1] > 1 else None, flags=ext.get("flags", 0), rgba=rgba, ids=ids,
                 want_rgba=rgba is not None, want_ids=ids is not None, want_t=False)
    return execute


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import distributed as D, scenes, types as T
    buf, frame, sky = scenes.random_spheres(W, H_, grid=6)
    ex = _oracle_execute(buf, sky)
    rgba, ids = D.render_tiles(ex, frame, W, H_, rank, world, "cpu", mode=T.MODE_REF)
    pt_tiles, _ = D.render_tiles(ex, frame, W, H_, rank, world, "cpu", want_ids=False, mode=T.MODE_PT, spp=2, max_bounces=3, seed=5)
    pt_samples = D.render_samples(ex, frame, W, H_, rank, world, "cpu", spp_per_rank=2, max_bounces=3, seed=5)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rgba=rgba.numpy(), ids=ids.numpy(), pt_tiles=pt_tiles.numpy(),
             pt_samples=pt_samples.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_tile_and_sample_sharding(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from raytracing_amd import scenes
    buf, frame, sky = scenes.random_spheres(W, H_, grid=6)
    ref = oracle.render(buf, frame, W, H_, sky=sky, mode=0, want_t=False)
    pt2 = oracle.render(buf, frame, W, H_, sky=sky, mode=1, spp=2, max_bounces=3, seed=5, want_ids=False, want_t=False)
    pt4 = oracle.render(buf, frame, W, H_, sky=sky, mode=1, spp=4, max_bounces=3, seed=5, want_ids=False, want_t=False)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        # tile sharding reassembles the one-process image exactly, on every rank
        assert np.array_equal(z["rgba"], ref["rgba"])
        assert np.array_equal(z["ids"], ref["ids"])
        assert np.array_equal(z["pt_tiles"], pt2["rgba"])
        # sample sharding: 2 ranks x 2 spp = the 4-spp image up to the order of additions
        assert np.allclose(z["pt_samples"][..., :3], pt4["rgba"][..., :3], rtol=1e-6, atol=1e-7)
        assert (z["pt_samples"][..., 3] == 1).all()


def test_shard_parameters():
    from raytracing_amd import distributed as D, types as T
    assert D.tile_shard_kwargs(0, 1) == {}
    assert D.tile_shard_kwargs(3, 8) == dict(shard_rank=3, shard_count=8)
    kw = D.sample_shard_kwargs(2, 4, 16)
    assert kw["sample_offset"] == 32 and kw["spp"] == 16 and kw["flags"] & T.FLAG_OUTPUT_SUM
    assert not (D.sample_shard_kwargs(0, 1, 16)["flags"] & T.FLAG_OUTPUT_SUM)
