"""CPU tests (gloo, world_size 2): the N>1 path of bench.py -- spp split + SUM reduce + resolve on rank 0.

The per-rank renders come from the oracle in COUNTER mode (same RNG keying as the HIP kernels), so the test pins the
property the multi-GPU path relies on: the union of the ranks' sample ranges reproduces the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracing_amd.parallel import row_range, sample_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ranges_tile_exactly():
    for world in (1, 2, 3, 4, 8):
        for total in (1, 7, 10, 1024, 2160):
            for fn in (sample_range, row_range):
                parts = [fn(r, world, total) for r in range(world)]
                assert parts[0][0] == 0 and parts[-1][1] == total
                assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
                sizes = [b - a for a, b in parts]
                assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sample_range(2, 2, 10)


def _worker(rank, world, port, spp, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from raytracing_amd.parallel import reduce_frame, sample_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    s = O.Scene.c0(48, 32, 1)
    k0, k1 = sample_range(rank, world, spp)
    part, _, cnt = s.render(k0, k1, 12, O.RNG_COUNTER, seed=4, n_threads=2, want_rgb=False)
    accum = torch.from_numpy(part.copy())
    reduce_frame(accum, world, dst=0)
    rays = torch.tensor([float(cnt[0])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), rays=rays.numpy(), rgb=O.resolve(accum.numpy(), spp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spp_split_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    import oracle_lib as O
    spp = 6
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(2, port, spp, out), nprocs=2, join=True)
    got = np.load(out)
    s = O.Scene.c0(48, 32, 1)
    want, want_rgb, cnt = s.render(0, spp, 12, O.RNG_COUNTER, seed=4, n_threads=2)
    assert np.allclose(got["accum"], want, rtol=1e-13, atol=0)      # only the fp64 summation order differs
    assert int(got["rays"][0]) == int(cnt[0])
    assert np.abs(got["rgb"].astype(int) - want_rgb.astype(int)).max() <= 1


def _gpu_worker(rank, world, port, spp, split, steps, out_path):
    """One rank of bench.py's N > 1 step with the HIP path: render this rank's share on the GPU (both ranks share device 0 on a one-GPU
    box), SUM-reduce onto rank 0 (gloo: through the host), resolve.  `steps` repeats the step as bench.py's timed loop does."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import raytracing_amd as R
    from raytracing_amd.host_scenes import HostScene
    from raytracing_amd.parallel import reduce_frame, row_range, sample_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w, h, depth = 64, 48, 12
    hs = HostScene.random_spheres(w, h, 1)
    sc = hs.upload()
    k0, k1 = sample_range(rank, world, spp) if split == "spp" else (0, spp)
    y0, y1 = (0, h) if split == "spp" else row_range(rank, world, h)
    accum = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    rays = 0
    for _ in range(steps):
        if split == "rows":
            accum[:y0].zero_(); accum[y1:].zero_()
        st = sc.render(accum.data_ptr(), w, h, k0, k1, depth, R.MODE_REFERENCE_FP64, seed=4, y_begin=y0, y_end=y1)
        host = accum.cpu()
        reduce_frame(host, world, dst=0)
        accum.copy_(host)
        rays = st["rays"]
    total = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(total)
    if rank == 0:
        np.savez(out_path, accum=accum.cpu().numpy(), rays=total.numpy())
    dist.barrier()
    sc.destroy(); hs.free()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("split,steps", [("spp", 1), ("rows", 2)])
def test_two_ranks_on_the_hip_path_match_one_rank(tmp_path, split, steps):
    """The N > 1 step of bench.py with the HIP kernels doing the rendering (2 ranks, gloo, one device): spp split -- fp64 sums equal the
    one-rank frame to the rounding of one addition per pixel; rows split over two steps -- the frame does not grow with the step count
    (a rank clears the rows it does not own before every step) and equals the one-rank frame bit for bit (disjoint rows: x + 0)."""
    import torch.multiprocessing as mp
    import raytracing_amd as R
    from raytracing_amd.host_scenes import HostScene
    spp = 6
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "frame.npz")
    mp.spawn(_gpu_worker, args=(2, port, spp, split, steps, out), nprocs=2, join=True)
    got = np.load(out)
    hs = HostScene.random_spheres(64, 48, 1)
    sc = hs.upload()
    want, _, st = sc.render_to_host(64, 48, 0, spp, 12, R.MODE_REFERENCE_FP64, seed=4)
    if split == "rows":
        assert np.array_equal(got["accum"].view(np.uint64), want.view(np.uint64))
    else:
        assert np.allclose(got["accum"], want, rtol=1e-13, atol=0)
    assert int(got["rays"][0]) == st["rays"]
    sc.destroy(); hs.free()


@pytest.mark.gpu
@pytest.mark.parametrize("mode_name", ["fp64ref", "fast64", "fast32"])
def test_frame_reduce_sums_partial_frames_of_several_scenes(mode_name):
    """rt_frame_reduce (one process, several devices; here three scene replicas on the one device there is): three partial frames over
    disjoint sample ranges, started without stats, collected with rt_render_wait, summed onto the first -- equal to the single render
    of all samples up to the rounding of two additions per value (fp64: 1e-13; fp32 fast mode: 1e-5, its tile sums add in completion
    order anyway), ray counts add up exactly."""
    import raytracing_amd as R
    from raytracing_amd.host_scenes import HostScene
    mode, dt, tol = {"fp64ref": (R.MODE_REFERENCE_FP64, np.float64, 1e-13), "fast64": (R.MODE_FAST_FP64, np.float64, 1e-13),
                     "fast32": (R.MODE_FAST_FP32, np.float32, 1e-5)}[mode_name]
    assert R.device_count() >= 1
    w, h, depth, spp = 80, 48, 12, 9
    hs = HostScene.random_spheres(w, h, 1)
    scenes = [hs.upload() for _ in range(3)]
    bufs = [R.DeviceBuffer(w * h * 3 * np.dtype(dt).itemsize) for _ in range(3)]
    ranges = [(0, 2), (2, 5), (5, 9)]
    for sc, b, (k0, k1) in zip(scenes, bufs, ranges):
        assert sc.render(b.ptr, w, h, k0, k1, depth, mode, seed=4, want_stats=False) is None
    stats = [sc.render_wait() for sc in scenes]
    with pytest.raises(R.RtError):
        scenes[0].render_wait()                    # nothing outstanding any more
    R.frame_reduce(bufs[0].ptr, 0, [bufs[1].ptr, bufs[2].ptr], [0, 0], dt == np.float64, w * h * 3)
    got = bufs[0].download(dt, (h, w, 3))
    want, _, st = scenes[0].render_to_host(w, h, 0, spp, depth, mode, seed=4)
    assert np.allclose(got, want, rtol=tol, atol=tol * float(np.abs(want).max()))
    assert sum(s["rays"] for s in stats) == st["rays"]
    R.frame_reduce(bufs[0].ptr, 0, [], [], dt == np.float64, w * h * 3)       # no sources: a no-op
    assert np.array_equal(bufs[0].download(dt, (h, w, 3)), got)
    for b in bufs:
        b.free()
    for sc in scenes:
        sc.destroy()
    hs.free()


@pytest.mark.gpu
def test_launches_with_more_than_48k_of_lds_on_every_scene_replica(monkeypatch, capfd):
    """More than 48 KiB of dynamic LDS needs hipFuncSetAttribute on the kernel -- per DEVICE, which the launcher's cache is keyed by
    (round 2 keyed it by thread: device 1 of a one-process-N-devices render never got the attribute once device 0 had).  RT_STACK_PAD
    reserves 40 stack levels beyond what the tree needs, so every workgroup asks for more than 48 KiB; two replicas of the scene (the
    one device twice: what a one-GPU box can rehearse of scene::render(..., {0, 1})) render halves of the samples without stats,
    rt_frame_reduce sums them, and the sum equals one render.  A second render of a scene whose first is outstanding is refused."""
    import re
    import raytracing_amd as R
    from raytracing_amd.host_scenes import HostScene
    monkeypatch.setenv("RT_STACK_PAD", "40")
    monkeypatch.setenv("RT_DEBUG", "1")
    w, h, depth, spp = 64, 48, 12, 4
    hs = HostScene.random_spheres(w, h, 1)
    scenes = [hs.upload() for _ in range(2)]
    err = capfd.readouterr().err
    tree_depth = max(int(m) for m in re.findall(r"fast tree: \d+ nodes, depth (\d+)", err))
    assert (tree_depth + 2 + 40) * 1024 + 768 * 4 + 16 > 48 * 1024
    for mode, dt in ((R.MODE_FAST_FP64, np.float64), (R.MODE_FAST_FP32, np.float32)):
        bufs = [R.DeviceBuffer(w * h * 3 * np.dtype(dt).itemsize) for _ in range(2)]
        for sc, b, (k0, k1) in zip(scenes, bufs, [(0, 2), (2, 4)]):
            sc.render(b.ptr, w, h, k0, k1, depth, mode, seed=9, want_stats=False)
        with pytest.raises(R.RtError):
            scenes[0].render(bufs[0].ptr, w, h, 0, 1, depth, mode, seed=9, want_stats=False)     # one set of counters per scene: refused while outstanding
        stats = [sc.render_wait() for sc in scenes]
        R.frame_reduce(bufs[0].ptr, 0, [bufs[1].ptr], [0], dt == np.float64, w * h * 3)
        got = bufs[0].download(dt, (h, w, 3))
        want, _, st = scenes[1].render_to_host(w, h, 0, spp, depth, mode, seed=9)
        tol = 1e-13 if dt == np.float64 else 1e-5
        assert np.allclose(got, want, rtol=tol, atol=tol * float(np.abs(want).max()))
        assert sum(x["rays"] for x in stats) == st["rays"] and st["rays"] > w * h * spp
        for b in bufs:
            b.free()
    R.frame_reduce_release()
    for sc in scenes:
        sc.destroy()
    hs.free()
