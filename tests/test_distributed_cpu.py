"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): sample slicing by global frame index + one sum
reduction of the accumulation buffers reproduces the single-process render. The CPU oracle stands in for the GPU
renderer here; the slicing / reduction code is the one bench.py uses."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_local, out_dir):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    from ray_tracer_amd import scenes
    from ray_tracer_amd.distributed import frame_slice, frames_of_rank, reduce_accumulators
    sc = scenes.cornell(32, 32).finalize()
    o = po.OracleScene(sc)
    first, stride = frame_slice(rank, world)
    assert frames_of_rank(rank, world, n_local) == [first + k * stride for k in range(n_local)]
    s, _, _ = o.render(n_local, 4, first=first, stride=stride, threads=2)
    t = torch.from_numpy(s.reshape(-1).copy())
    reduce_accumulators(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy().reshape(32, 32, 4))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_slices_reduce_to_the_single_process_image(tmp_path, oracle):
    import torch.multiprocessing as mp
    from ray_tracer_amd import scenes
    world, n_local = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), n_local, str(tmp_path)), nprocs=world, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    sc = scenes.cornell(32, 32).finalize()
    full, _, _ = oracle.OracleScene(sc).render(world * n_local, 4)
    assert (reduced[..., 3] == world * n_local).all()
    np.testing.assert_allclose(reduced, full, rtol=1e-5, atol=1e-6)   # differs only by fp32 summation order


def test_frame_slice_validation():
    from ray_tracer_amd.distributed import frame_slice, frames_of_rank, mean_from_sum
    assert frame_slice(3, 8) == (3, 8)
    assert frames_of_rank(1, 4, 3) == [1, 5, 9]
    with pytest.raises(ValueError):
        frame_slice(4, 4)
    covered = sorted(f for r in range(4) for f in frames_of_rank(r, 4, 5))
    assert covered == list(range(20))
    m = mean_from_sum(np.array([[[2.0, 4.0, 6.0, 2.0]]], np.float32))
    assert m.tolist() == [[[1.0, 2.0, 3.0]]]


@pytest.mark.parametrize("w,h", [(64, 32), (100, 52), (7, 9), (1, 1), (17, 8)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_tile_slices_partition_the_image(w, h, world):
    """The 8x8 tiles dealt round-robin to the ranks cover every pixel exactly once, whatever the image size."""
    from ray_tracer_amd.distributed import tile_mask, tile_slice
    owners = np.zeros((h, w), np.int32)
    for rank in range(world):
        m = tile_mask(w, h, *tile_slice(rank, world))
        assert m.shape == (h, w) and m.dtype == bool
        owners += m
        # whole tiles: every 8x8 block is owned entirely or not at all
        for y0 in range(0, h, 8):
            for x0 in range(0, w, 8):
                blk = m[y0:y0 + 8, x0:x0 + 8]
                assert blk.all() or not blk.any()
    assert (owners == 1).all()
    with pytest.raises(ValueError):
        tile_slice(world, world)
