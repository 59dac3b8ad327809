"""The multi-GPU path of BASELINE configs[4], as far as one GPU allows: the product renderer (not the oracle) slices samples
by global frame index, and the fp32 accumulation buffers are sum-reduced in place through the zero-copy torch view of the
library's device buffer, with the same reduce_accumulators call bench.py uses."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_worker(rank, port):
    # torch first: its HIP runtime must be the one the process initialises (bench.py imports in the same order)
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import ray_tracer_amd as rt
    from ray_tracer_amd import scenes
    from ray_tracer_amd.distributed import accumulator_tensor, frame_slice, reduce_accumulators
    sc = scenes.cornell(128, 96).finalize()
    r = rt.HipRenderer.create(sc, max_bounces=4)
    r.set_frame_slice(*frame_slice(0, 1))
    r.render(3)
    before = r.read_sum()
    assert before[..., 3].min() == 3 and before[..., :3].max() > 0
    t = accumulator_tensor(r, torch.device("cuda", 0))
    ptr, n_bytes = r.accum_device_buffer()
    assert t.data_ptr() == ptr and t.numel() * 4 == n_bytes == 128 * 96 * 16 and t.dtype == torch.float32
    assert np.array_equal(t.cpu().numpy().reshape(96, 128, 4), before)          # the view sees the library's data
    reduce_accumulators(t)                                                       # RCCL all-reduce, in place
    torch.cuda.synchronize()
    assert np.array_equal(r.read_sum(), before)                                  # sum over one rank
    t.mul_(2.0)                                                                  # writes through the view land in the library's buffer
    torch.cuda.synchronize()
    assert np.array_equal(r.read_sum(), before * 2.0)
    r.destroy()
    dist.destroy_process_group()


def test_rccl_all_reduce_of_the_zero_copy_accumulator_view():
    """World size 1 over the `nccl` backend (RCCL): the collective runs in place on the library's own device buffer, and
    what rt_read_sum_f32 returns afterwards is the reduced buffer. Runs in a child process so that torch initialises the
    GPU runtime before the render library does, as in bench.py."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(_free_port(),), nprocs=1, join=True)


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
    import ray_tracer_amd as rt
    from ray_tracer_amd import scenes
    from ray_tracer_amd.distributed import accumulator_tensor, frame_slice, reduce_accumulators
    sc = scenes.atrium(width=320, height=180, scale=0.05).finalize(spatial=True)
    r = rt.HipRenderer.create(sc, max_bounces=8)          # both ranks share GPU 0 here; on a node each has its own
    r.set_frame_slice(*frame_slice(rank, world))
    r.render(n_local)
    t = accumulator_tensor(r, torch.device("cuda", 0))    # zero-copy view of the library's sum buffer
    staged = t.cpu()                                       # gloo reduces host tensors; RCCL reduces `t` itself
    reduce_accumulators(staged)
    t.copy_(staged)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), r.read_sum())
    dist.barrier()
    r.destroy()
    dist.destroy_process_group()


def test_two_ranks_of_the_product_renderer_reduce_to_the_single_context_image(rt, tmp_path):
    """Two processes, each a HipRenderer rendering its slice of the global frame indices; after the reduction rank 0's
    accumulation buffer equals one context rendering all frames (fp32 summation order aside)."""
    import torch.multiprocessing as mp
    from ray_tracer_amd import scenes
    world, n_local = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), n_local, str(tmp_path)), nprocs=world, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    sc = scenes.atrium(width=320, height=180, scale=0.05).finalize(spatial=True)
    whole = rt.HipRenderer.create(sc, max_bounces=8)
    whole.render(world * n_local)
    ref = whole.read_sum()
    whole.destroy()
    assert (reduced[..., 3] == world * n_local).all()
    np.testing.assert_allclose(reduced, ref, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("launcher", ["torch.distributed.run", "none"])
@pytest.mark.parametrize("partition,label,scaling,rays_per_pixel_step", [("samples", "sample-slice x2", "weak", 2), ("tiles", "tile-slice x2", "strong", 1)])
def test_bench_line_from_two_ranks_on_one_gpu(partition, label, scaling, rays_per_pixel_step, launcher):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process), with both ranks on GPU 0
    and the collectives over gloo (RT_BENCH_REHEARSE_ONE_GPU): rank 0 alone prints one JSON line, it carries the contract's
    keys, counts the samples of both ranks and says that it is a rehearsal. launcher "none": plain `python bench.py --gpus 2`,
    which has to start its two ranks by itself and print the same single line."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_REHEARSE_ONE_GPU"] = "1"
    steps, warmup = 4, 2
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port())] if launcher != "none" else [sys.executable]
    cmd = head + [str(root / "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
                  "--scale", "0.05", "--width", "320", "--height", "180", "--partition", partition]
    done = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                 # one line, from rank 0 only
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == scaling
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["config"]["parallelism"] == label and "rehearsal" in d
    assert d["collective"]["ranks"] == 2 and d["collective"]["partition"] == partition
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # both ranks' work is in the aggregate: two camera rays per pixel and step when the ranks take different frames, one
    # when they share the frames tile by tile
    assert d["extend_rays"] >= rays_per_pixel_step * steps * 320 * 180
    if partition == "tiles":
        assert d["extend_rays"] < 2 * steps * 320 * 180 * 4
    assert "cpu_baseline" not in d and "one_eighth_measured" not in d          # single-GPU legs stay out of an N > 1 line
