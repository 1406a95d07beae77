#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X path-tracing core on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment); started WITHOUT a launcher, bench.py starts those N ranks itself as a child process -- before it makes
any GPU call -- relays rank 0's one JSON line and exits with the launcher's return code.

A *step* is one pass of the hot path over one frame: every rank renders its share of the frame's samples with the
HIP kernels (rt_render), the per-rank accumulation buffers are summed onto rank 0 by an RCCL reduce over xGMI,
and rank 0 resolves the frame (mean, gamma, clamp, 8-bit).  The samples of the frame are split over the ranks
(spp split; RNG streams are keyed by the GLOBAL sample index, so the image does not depend on N): total work is
fixed as N grows => "scaling": "strong".

Workload at N=1 (config.workload): BASELINE configs[2], the config the metric is quoted on ("4K 1024spp OBJ scene") --
a ~1.3 M-triangle OBJ mesh (procedural: no assets ship with the reference and none can be fetched) read through the
OBJ loader of the C++ host mirror, reference BVH builder, 3840x2160, 1024 spp, max_depth 50.  `--workload
random_spheres` runs configs[1] instead: the reference's random_scene() (~485 spheres) + the documented area light.
Rays are counted by the kernel: trace() invocations that traverse the scene (raytracing.cpp:32-37).

Precision: the default mode is RT_MODE_FAST_FP64 -- the reference's arithmetic (fp64, its evaluation order) for rays, primitive
tests, shading and sums, behind the library's SAH tree whose boxes only cull (dtype "f64": the reference computes in fp64).
`--mode fp32` is the fp32 fast mode; at N=1 its rate on the same frame is reported next to the headline as `fp32_mode`.

The one JSON line also carries
  roofline      the render kernel against the roof that binds it (VALU issue; frac <= 1 by construction), plus the vector-L1 and HBM
                fractions: algorithmic work per launch / HIP-event duration measured in this run;
  image         the timed scene at reduced resolution: GPU vs the CPU oracle on the same streams (rel. RMSE, PSNR) and converged
                (independent streams) against the CPU-vs-CPU noise floor;
  cpu_reference (extra) the unmodified reference binary oracle/_ref/ref_harness, one thread, bounded sample; absent if not built
  cpu_baseline  the oracle (CPU restatement of the reference, fp64) timed on this box's host cores on a bounded
                sample of the same workload -- a reported baseline, not the target.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic cost model of the dominant kernel (DESIGN.md "Measurement"): per box test 6 fma + 6 min/max + 4 = 22 flop
# (fma = 2) and half a node record; per leaf test ~40 flop (triangle: 2 cross, 4 dot, 1 rcp; sphere: 3 dots, disc, sqrt,
# div) and one primitive record; per ray ~150 flop of shading/sampling/regeneration.  Box tests are fp32 in every fast mode;
# leaf tests and shading are fp64 in the fp64 modes.
FLOP_BOX, FLOP_LEAF, FLOP_SHADE = 24.0, 40.0, 150.0
PEAK_FP32_TFLOPS = 157.3     # MI355X vector fp32 peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP64_TFLOPS = 78.6      # vector fp64
PEAK_HBM_GBS = 8000.0
PEAK_L1_GBS = 256 * 64 * 2.4   # vector L1 (TCP) -> register path: 64 B/clk/CU x 256 CUs x 2.4 GHz = 39.3 TB/s

MODES = {"fp64": "MODE_FAST_FP64", "fp32": "MODE_FAST_FP32", "fp32ref": "MODE_FP32_REFERENCE_TREE", "fp64ref": "MODE_REFERENCE_FP64"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel of the frame (split over the ranks)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--mode", choices=list(MODES), default="fp64",
                    help="fp64: fast SAH tree, fp64 results (the reference's precision; default); fp32: fast tree in fp32; fp32ref: reference tree in fp32; "
                         "fp64ref: reference tree and visit order in fp64 (the bit-exact parity mode)")
    ap.add_argument("--no-fp32", action="store_true", help="N=1: skip the extra fp32-mode timing of the same frame")
    ap.add_argument("--image-width", type=int, default=240, help="width of the reduced-resolution image check (same aspect, 0 = skip)")
    ap.add_argument("--image-spp", type=int, default=256, help="spp of the converged (independent streams) image check")
    ap.add_argument("--workload", choices=["obj", "random_spheres", "soup"], default="obj", help="obj: BASELINE configs[2] (the config the metric is quoted on); random_spheres: configs[1]; soup: configs[3] (triangle soup with image textures)")
    ap.add_argument("--soup-tris", type=int, default=10_000_000)
    ap.add_argument("--soup-tex", type=int, default=1024, help="edge of each of the 64 map_Kd textures")
    ap.add_argument("--obj", default="auto", help="OBJ file of the obj workload; auto = generate the procedural mesh (raytracing_amd/tools/make_mesh.py)")
    ap.add_argument("--mesh-level", type=int, default=8, help="icosphere subdivision of the generated mesh: 8 -> 1.31 M triangles")
    ap.add_argument("--split", choices=["spp", "rows"], default="spp", help="N > 1: split the frame's samples (default) or its rows over the ranks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the cpu_baseline leg (full resolution): 16 = about 12 s of work on 16 cores for the mesh workload")
    return ap.parse_args()


def oracle_scene(args, host_arrays):
    """The CPU oracle's scene for this workload (tests/oracle_lib.py over oracle/librt_oracle.so; checker and baseline only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # random_spheres: the oracle's own builder (nested bvh over shutter [0,1], as the reference builds it); OBJ: static
    # triangles re-created from the flattened arrays
    return O, (O.Scene.c0(args.width, args.height, 1) if args.workload == "random_spheres" else O.scene_from_arrays(host_arrays, 1))


def cpu_baseline(args, O, s):
    """Oracle (kind 'port': CPU restatement of the reference, pinned bit-exact to it by tests/golden) on the host cores.
    Bounded sample: the same scene, resolution and depth at --cpu-spp samples per pixel."""
    cores = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    _, _, cnt = s.render(0, args.cpu_spp, args.depth, O.RNG_COUNTER, seed=1, n_threads=cores, want_rgb=False)
    dt = time.perf_counter() - t0
    return {"value": round(int(cnt[0]) / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{args.workload} {args.width}x{args.height}, {args.cpu_spp} spp, max_depth {args.depth}: "
                      f"{int(cnt[0])} rays in {dt:.2f} s (fp64, reference tree, OpenMP over rows, per-(pixel,sample) RNG streams)"}


def image_check(args, R, O, s, scene, mode):
    """The metric's second half ("image RMSE"): the TIMED scene at reduced resolution (same aspect, same depth), GPU against the CPU
    oracle.  (1) same streams, 16 spp: every sample draws the same numbers on both sides, so the difference is the arithmetic's;
    (2) converged, independent streams (different seeds) at --image-spp: GPU-vs-CPU relative RMSE next to the CPU-vs-CPU RMSE of two
    seeds, the Monte-Carlo noise floor at that spp.  Relative RMSE = rms(diff of mean linear radiance) / rms(CPU radiance); PSNR on
    the 8-bit images."""
    import numpy as np
    w = args.image_width
    h = max(2, round(w * args.height / args.width))
    s.set_size(w, h)
    cores = min(16, os.cpu_count() or 1)

    def psnr(x, y):
        mse = float(np.mean((x.astype(np.float64) - y.astype(np.float64)) ** 2))
        return None if mse == 0 else round(10 * math.log10(255.0 ** 2 / mse), 2)

    def rel(x, y, spp):
        return float(np.sqrt(np.mean((x - y) ** 2)) / max(np.sqrt(np.mean(y ** 2)), 1e-300))

    spp_same = 16
    cpu, cpu_rgb, cnt = s.render(0, spp_same, args.depth, O.RNG_COUNTER, seed=1, n_threads=cores)
    gpu, gpu_rgb, st = scene.render_to_host(w, h, 0, spp_same, args.depth, mode, seed=1)
    out = {"width": w, "height": h, "max_depth": args.depth,
           "same_streams": {"spp": spp_same, "rel_rmse": float(f"{rel(gpu.astype(np.float64), cpu, spp_same):.3e}"), "psnr_8bit_dB": psnr(gpu_rgb, cpu_rgb),
                            "psnr_note": "null = the 8-bit images are identical", "rays_gpu": st["rays"], "rays_cpu": int(cnt[0])}}
    spp = args.image_spp
    if spp > 0:
        a, a_rgb, _ = s.render(0, spp, args.depth, O.RNG_COUNTER, seed=101, n_threads=cores)
        b, b_rgb, _ = s.render(0, spp, args.depth, O.RNG_COUNTER, seed=202, n_threads=cores)
        c, c_rgb, _ = scene.render_to_host(w, h, 0, spp, args.depth, mode, seed=303)
        out["converged"] = {"spp": spp, "rel_rmse_gpu_vs_cpu": round(rel(c.astype(np.float64), a, spp), 4), "rel_rmse_cpu_vs_cpu": round(rel(b, a, spp), 4),
                            "psnr_gpu_vs_cpu_dB": psnr(c_rgb, a_rgb), "psnr_cpu_vs_cpu_dB": psnr(b_rgb, a_rgb),
                            "tolerance": "rel_rmse_gpu_vs_cpu <= 1.25 x rel_rmse_cpu_vs_cpu (the noise floor); PSNR within 1 dB of the CPU-vs-CPU PSNR"}
    s.set_size(args.width, args.height)
    return out


def cpu_reference(args, obj_path):
    """The REAL reference (oracle/_ref/ref_harness: /root/reference's own sources, built in the dev container by `make -C oracle ref`;
    the binary travels with the tree, the sources do not) on one host thread -- its best configuration: its OpenMP loop serialises on
    rand()'s lock.  Bounded sample: the workload's scene at 1/3 of the resolution, REF_SPP samples per pixel (about 8 s).  Returns None when the binary is not there."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(exe):
        return None
    w, h = max(args.width // 3, 16), max(args.height // 3, 16)
    ref_spp = 1 if args.workload == "soup" else REF_SPP          # a soup ray costs the CPU tree hundreds of box tests
    with tempfile.TemporaryDirectory() as tmp:
        if args.workload == "random_spheres":
            cmd = [exe, tmp, "time_c0", "1", str(w), str(h), str(ref_spp), str(args.depth)]
        else:
            cmd = [exe, tmp, "time_obj", obj_path, str(w), str(h), str(ref_spp), str(args.depth)] + [str(v) for v in OBJ_CAMERA[args.workload]]
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
        except (OSError, subprocess.TimeoutExpired):
            return None
    rays = re.search(r"image \d+x\d+ spp=\d+ depth=\d+ rays=(\d+)", res.stdout)
    secs = re.search(r"INFO time_\w+ seconds=([0-9.]+)", res.stdout)
    if res.returncode != 0 or not rays or not secs:
        return None
    n, dt = int(rays.group(1)), float(secs.group(1))
    return {"value": round(n / dt / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "reference",
            "sample": f"{args.workload} {w}x{h}, {ref_spp} spp, max_depth {args.depth}: {n} rays in {dt:.2f} s (the reference's own trace() loop, single thread, rand())"}


REF_SPP = 8        # samples per pixel of the cpu_reference leg
OBJ_CAMERA = {"obj": (0, 1.8, 5.0, 0, 1.0, 0, 40.0), "soup": (0, 5.0, 75.0, 0, 0, 0, 40.0)}     # position, look-at, fov of the OBJ workloads


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (python -m torch.distributed.run, one per GPU, rendezvous on
    127.0.0.1) as a CHILD process before this process has made any GPU call, relay rank 0's one JSON line and exit with the launcher's
    return code.  (Never an exec: a process that has touched the GPU must not be replaced.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)          # stderr passes through
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    sys.exit(res.returncode if res.returncode != 0 or lines else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                     # does not return
    if args.workload == "soup":
        args.cpu_spp = 1                  # bounded sample: a soup ray costs the CPU tree hundreds of box tests
    import torch
    import torch.distributed as dist
    import raytracing_amd as R
    from raytracing_amd.host_scenes import HostScene
    from raytracing_amd.parallel import reduce_frame, row_range, sample_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                 # the launcher decides
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path is the only path")
    # one rank per GPU (RCCL).  RT_BENCH_BACKEND=gloo is for rehearsing the N > 1 path on a box with fewer GPUs than ranks:
    # ranks then share devices and the reduce goes through the host
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    # RT_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): a process group of one, every collective of the N > 1 path issued
    # all the same -- what a one-GPU box can rehearse of the RCCL branch
    grouped = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST", "0") == "1"
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    # stdout carries the JSON line and nothing else: the host mirror prints the reference's "[INFO] building bvh..." lines there and RCCL
    # its version banner, so file descriptor 1 points at stderr until the line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    w, h, spp = args.width, args.height, args.spp
    mode = getattr(R, MODES[args.mode])
    is_fp64 = args.mode in ("fp64", "fp64ref")
    dtype = torch.float64 if is_fp64 else torch.float32
    # The scene is built ONCE per node: rank 0 writes the OBJ, runs it through the OBJ loader and the reference BVH builder of the host
    # mirror and -- for N > 1 -- saves the flattened description with rt_scene_desc_save (the binary scene cache); the
    # other ranks wait at a barrier and load that file instead of repeating eight 100 MB OBJ writes, parses and tree builds.
    obj_path, host = None, None
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_bench_scene_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}.rtscene")
    if rank == 0:
        if args.workload == "random_spheres":
            host = HostScene.random_spheres(w, h, 1)
        elif args.workload == "soup":
            from raytracing_amd.tools.make_mesh import write_soup
            obj_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_soup_{args.soup_tris}_{os.getpid()}.obj")
            write_soup(obj_path, args.soup_tris, 64, args.soup_tex)
            host = HostScene.obj(obj_path, w, h, 1, cam_pos=OBJ_CAMERA["soup"][0:3], cam_lookat=OBJ_CAMERA["soup"][3:6], fov=OBJ_CAMERA["soup"][6])
        else:
            obj_path = args.obj
            if obj_path == "auto":
                from raytracing_amd.tools.make_mesh import write_scene
                obj_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_blob_l{args.mesh_level}_{os.getpid()}.obj")
                write_scene(obj_path, args.mesh_level)
            host = HostScene.obj(obj_path, w, h, 1, cam_pos=OBJ_CAMERA["obj"][0:3], cam_lookat=OBJ_CAMERA["obj"][3:6], fov=OBJ_CAMERA["obj"][6])
        n_prims = int(host.desc.contents.n_prims)
        if world > 1:
            from raytracing_amd.rtcore import _check, load_library
            _check(load_library().rt_scene_desc_save(host.desc, os.fsencode(cache + ".tmp")))
            os.replace(cache + ".tmp", cache)
        scene = host.upload()
    if world > 1:
        dist.barrier()                      # the cache file is complete
        if rank != 0:
            arrays = R.load_scene_cache(cache)
            n_prims = len(arrays["prim_kind"])
            scene = R.Scene(arrays)
            del arrays
        dist.barrier()                      # everyone has read it
        if rank == 0:
            os.unlink(cache)
    if args.workload == "random_spheres":
        workload = f"BASELINE configs[1]: random_scene (reference demo.hpp:17-60, rand() seed 1) + area light, {w}x{h}, {spp} spp, max_depth {args.depth}"
    elif args.workload == "soup":
        workload = (f"BASELINE configs[3]: {n_prims}-triangle soup in [-50,50]^3, 64 MTL materials each with a {args.soup_tex}^2 PNG map_Kd, emissive quad, "
                    f"through the OBJ loader and the reference BVH builder, {w}x{h}, {spp} spp, max_depth {args.depth}")
    else:
        workload = (f"BASELINE configs[2]: procedural OBJ mesh ({n_prims} triangles, diffuse/specular MTL + emissive quad) through the OBJ loader "
                    f"and the reference BVH builder, {w}x{h}, {spp} spp, max_depth {args.depth}")

    # this rank's share of the frame's samples
    k0, k1 = sample_range(rank, world, spp) if args.split == "spp" else (0, spp)
    y0, y1 = (0, h) if args.split == "spp" else row_range(rank, world, h)     # rows split: the other rows of `accum` stay zero, the reduce gathers
    accum = torch.zeros((h, w, 3), dtype=dtype, device="cuda")        # rows split: the rows of other ranks stay zero, the reduce gathers
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(step_mode=mode, buf=None):
        buf = accum if buf is None else buf
        if args.split == "rows" and world > 1:
            # rows mode: a rank's buffer is zero outside its rows, so the SUM reduce is a gather.  Rank 0's other rows hold the last step's
            # reduced frame: clear them (render overwrites [y0, y1) itself)
            buf[:y0].zero_(); buf[y1:].zero_()
        st = scene.render(buf.data_ptr(), w, h, k0, k1, args.depth, step_mode, seed=1, stream=stream, y_begin=y0, y_end=y1)   # syncs, returns kernel stats
        reduce_frame(buf, world, dst=0, force=grouped)
        if rank == 0:
            R.resolve(buf.data_ptr(), buf.dtype == torch.float64, w * h, spp, rgb.data_ptr(), stream=stream)
        return st

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0            # trace() calls past the depth check: what the reference counts as a ray
    untraced = 0        # of those, the zero-direction segments the kernel finishes without walking the tree (rt_render_stats.untraced_rays)
    kernel_ms = []
    for _ in range(args.steps):
        st = step()
        rays += st["rays"]
        untraced += st["untraced_rays"]
        kernel_ms.append(st["kernel_ms"])
    fence()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(rays), float(untraced), elapsed], dtype=torch.float64, device="cuda")
    if grouped:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[2])
    reference_rays = float(tot[0])                  # all trace() calls of all ranks
    total_rays = float(tot[0]) - float(tot[1])      # the metric: rays that underwent a BVH traversal

    if rank == 0:
        # test counts for the roofline: one diagnostic launch (counting kernel variant) on a 1-spp sample of this rank's work
        probe = scene.render(accum.data_ptr(), w, h, k0, k0 + 1, args.depth, mode, seed=1, stream=stream, flags=R.FLAG_COUNT_TESTS, y_begin=y0, y_end=y1)
        probe_traced = max(probe["rays"] - probe["untraced_rays"], 1)
        box_per_ray = probe["box_tests"] / probe_traced           # per TRACED ray: an untraced segment costs no test and is charged no flop
        leaf_per_ray = probe["leaf_tests"] / probe_traced
        zero_share = probe["zero_direction_rays"] / max(probe["rays"], 1)
        rays_per_launch = (rays - untraced) / max(args.steps, 1)
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        secs = avg_ms * 1e-3
        fast = args.mode in ("fp64", "fp32")
        # flops by precision: box tests are fp32 in the fast modes; leaf tests and shading are in the mode's result precision
        flop32 = box_per_ray * FLOP_BOX if (fast or not is_fp64) else 0.0
        flop_res = leaf_per_ray * FLOP_LEAF + FLOP_SHADE + (0.0 if (fast or not is_fp64) else box_per_ray * FLOP_BOX)
        if is_fp64:
            flop64 = flop_res
        else:
            flop32, flop64 = flop32 + flop_res, 0.0
        flop_per_ray = flop32 + flop64
        achieved_tflops = rays_per_launch * flop_per_ray / secs / 1e12
        # the VALU roof for THIS mix of fp32 and fp64 work: total flops / the time the mix takes at the two vector peaks
        time_at_peak = rays_per_launch * (flop32 / (PEAK_FP32_TFLOPS * 1e12) + flop64 / (PEAK_FP64_TFLOPS * 1e12))
        peak_mix = rays_per_launch * flop_per_ray / time_at_peak / 1e12
        valu_frac = time_at_peak / secs
        # OBJ-loader scenes (triangles + phong only) walk the quantised four-wide nodes by default (rtcore.hip, RT_QNODES overrides)
        kernel_name = scene.kernel_name(mode)
        # launches of at most 32 samples per pixel of a triangle scene take the one-wave-workgroup twin of that kernel (rt_wave.h render_wave1_*;
        # RT_ONE_WAVE_SPP moves the limit): name the kernel this run's launches used
        if k1 - k0 <= int(os.environ.get("RT_ONE_WAVE_SPP", "32")) and kernel_name.rstrip(">").endswith(" 3") and "render_wave_" in kernel_name:
            kernel_name = kernel_name.replace("render_wave_", "render_wave1_")
        # render_wave_kernel<R, count, wide nodes, scene class>; render_wave_sorted_kernel<R, count, scene class> always walks the wide nodes
        quantised = fast and ("render_wave_sorted_kernel" in kernel_name or kernel_name.split(",")[2].strip() == "true")
        # a fast-mode node record is 64 B and holds its children's boxes: four quantised ones or two fp32 ones
        node_bytes = 64.0 if fast else (56.0 if is_fp64 else 32.0)
        boxes_per_node = (4 if quantised else 2) if fast else 1
        leaf_bytes = 96.0 if is_fp64 else 48.0
        bytes_per_ray = box_per_ray * node_bytes / boxes_per_node + leaf_per_ray * leaf_bytes
        fetch_gbs = rays_per_launch * bytes_per_ray / secs / 1e9
        scene_mb = (n_prims * node_bytes * 1.2 / boxes_per_node + n_prims * leaf_bytes) / 1e6      # ~0.6 binary SAH nodes per primitive (half as many wide ones) + primitive records
        # HBM-side bytes per launch: bytes per ray measured with rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes,
        # FETCH_SIZE doubled per MI355X_MICROARCH.md) on this workload and mode at reduced spp (profiles/hbm_traffic.json, written by
        # tools/hbm_traffic.py from a run of this same command under the profiler), scaled by this launch's ray count; null when no
        # measurement of this workload and mode is on file.  A counter pass cannot run inside the timed run.
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                rec = json.load(f).get(f"{args.workload}:{n_prims}:{args.mode}:{w}x{h}:d{args.depth}")
            if rec:
                per_launch = rays_per_launch if rec.get("ray_count") == "traced" else rays / max(args.steps, 1)     # older records are per trace() call
                traffic = int(rec["bytes_per_ray"] * per_launch)
                traffic_src = f"profiles/hbm_traffic.json: {rec['bytes_per_ray']:.1f} B/ray measured at {rec.get('spp_profiled')} spp ({rec.get('round', 'r01')}), x this launch's rays"
        except (OSError, ValueError):
            pass
        hbm_frac = round(traffic / secs / 1e9 / PEAK_HBM_GBS, 4) if traffic is not None else None
        # unit utilisation of the SAME kernel variant (rt_render_kernel_name: the kernels are compiled per scene class and node format) from
        # the PMC passes on file for this workload and mode (tools/collect_profiles.py): the latest round's, the longest launch's.  Stored,
        # not measured by this run -- the line says so.  A file whose kernel name differs (an older kernel) is not used.
        pmc = None
        try:
            import glob
            import re
            files = glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_wave_kernel_{args.mode}_{args.workload}_4k_*spp.json"))
            key = lambda f: (re.match(r"(r\d+_[a-z]+)_", os.path.basename(f)).group(1), int(re.search(r"_(\d+)spp\.json$", f).group(1)))
            for f in sorted(files, key=key, reverse=True) if (w, h) == (3840, 2160) else []:
                with open(f) as fh:
                    derived = json.load(fh).get("derived", {})
                hit = [v for name, v in derived.items() if kernel_name + "(" in name.replace("void ", "")]
                if hit:
                    v = hit[0]
                    pmc = {"valu_busy": round(v["valu_busy"], 3), "ta_busy": round(v["ta_busy"], 3), "waves_per_simd": round(v.get("waves_per_simd", 0.0), 2),
                           "valu_lane_slots_per_ray": v.get("valu_lane_slots_per_ray"), "wait_any_share_of_wave_cycles": v.get("wait_any_share"),
                           "source": f"profiles/{os.path.basename(f)}: rocprofv3 --pmc passes of this command at {key(f)[1]} spp, kernel {kernel_name}; not measured by this run"}
                    break
        except (OSError, ValueError, KeyError, AttributeError):
            pass
        assert 0.0 < valu_frac <= 1.0, f"roofline fraction {valu_frac} outside (0, 1]: the cost model or the peak is wrong"
        roofline = {"bound": "valu", "achieved": round(achieved_tflops, 3), "peak": round(peak_mix, 1), "unit": "TFLOP/s", "frac": round(valu_frac, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "hbm_frac": hbm_frac, "hbm_peak_GBs": PEAK_HBM_GBS, "pmc": pmc,
                    "l1_frac": round(fetch_gbs / PEAK_L1_GBS, 4), "l1_achieved_GBs": round(fetch_gbs, 1), "l1_peak_GBs": round(PEAK_L1_GBS, 0),
                    "kernel": kernel_name, "kernel_ms_avg": round(avg_ms, 3), "rays_per_launch": int(rays_per_launch),
                    "box_tests_per_ray": round(box_per_ray, 2), "leaf_tests_per_ray": round(leaf_per_ray, 2),
                    "flop_per_ray": {"fp32": round(flop32, 1), "fp64": round(flop64, 1)}, "record_bytes_per_ray": round(bytes_per_ray, 1),
                    "node_record_bytes": node_bytes, "boxes_per_node_record": boxes_per_node, "leaf_record_bytes": leaf_bytes, "scene_MB": round(scene_mb, 1),
                    "note": "No dense contraction (no MFMA) and no HBM stream: node and primitive records are served by L1/L2/Infinity Cache (hbm_frac = measured "
                            "memory-side bytes / time / 8 TB/s).  The nearest roof is VALU issue under wave64 divergence: frac = algorithmic flops per ray that walked "
                            "the tree (24 per fp32 box test, 40 per leaf test, 150 per ray of shading; leaf + shading in the result precision) x those rays / "
                            "kernel time, against the vector peaks 157.3 (fp32) / 78.6 (fp64) TFLOP/s weighted by this mix (`peak`).  l1_frac = algorithmic record "
                            "bytes / time against the vector-L1 peak.  `pmc` (stored passes of this command, same kernel variant): share of cycles the vector ALUs "
                            "and the texture-address units are busy, resident waves per SIMD, issued VALU lane slots per ray, share of wave cycles spent waiting"}
        out = {
            "metric": "Mrays/s (node), path-traced frame, results in the reference's precision; image RMSE vs the reference CPU path in `image`",
            "value": round(total_rays / elapsed / 1e6, 2),            # rays that walked the tree / wall time
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64" if is_fp64 else "f32",
            "data": "synthetic (procedural scene: random_scene() with rand() seed 1, or the generated OBJ mesh)",
            "config": {"workload": workload, "width": w, "height": h, "spp": spp, "max_depth": args.depth, "mode": MODES[args.mode].replace("MODE_", "RT_MODE_"),
                       "parallelism": f"{args.split} split over {world} rank(s), {'RCCL' if backend == 'nccl' else backend + ' (rehearsal: ranks may share a GPU)'} reduce of the accumulation buffer" if world > 1 else ("single GPU" if not grouped else f"single GPU, {backend} process group of one (every collective of the N > 1 path issued)"),
                       "rays_per_step": int(total_rays / max(args.steps, 1)), "samples_per_step": w * h * spp, "n_primitives": n_prims,
                       "reference_rays_per_step": int(reference_rays / max(args.steps, 1)),
                       "zero_direction_ray_share": round(zero_share, 4),
                       "ray_count_note": "`value` and rays_per_step count the rays that walked the tree.  The reference counts more (reference_rays_per_step, "
                                         "`value_reference_ray_count`): phong leaves wo unset (material.hpp:225-229,252-254), so a share of its trace() calls "
                                         "(zero_direction_ray_share) carries direction (0,0,0); such a ray can only miss in a scene of triangles "
                                         "(polygon.hpp:95-97) and the fast kernels finish it without a traversal (same result, same sample counts)"},
            "value_reference_ray_count": round(reference_rays / elapsed / 1e6, 2),
            "roofline": roofline,
        }
        if world == 1 and not args.no_fp32 and args.mode == "fp64":
            # the same frame in the fp32 fast mode (results held to an RMSE bar instead of bit parity): one untimed + one timed step
            acc32 = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
            step(R.MODE_FAST_FP32, acc32)
            fence()
            t1 = time.perf_counter()
            st32 = step(R.MODE_FAST_FP32, acc32)
            fence()
            dt32 = time.perf_counter() - t1
            out["fp32_mode"] = {"value": round((st32["rays"] - st32["untraced_rays"]) / dt32 / 1e6, 2), "unit": "Mrays/s", "dtype": "f32", "mode": "RT_MODE_FAST_FP32", "steps": 1,
                                "ms_per_step": round(dt32 * 1e3, 3), "kernel_ms": round(st32["kernel_ms"], 3), "rays_per_step": st32["rays"] - st32["untraced_rays"],
                                "value_reference_ray_count": round(st32["rays"] / dt32 / 1e6, 2)}
            del acc32
        if world == 1 and not args.no_cpu:
            O, osc = oracle_scene(args, host.arrays())
            out["cpu_baseline"] = cpu_baseline(args, O, osc)
            if args.image_width > 0:
                out["image"] = image_check(args, R, O, osc, scene, mode)
            osc.free()
            ref = cpu_reference(args, obj_path if args.workload != "random_spheres" else None)
            if ref:
                out["cpu_reference"] = ref      # extra to the contract: the unmodified reference next to the multi-threaded port
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    scene.destroy()
    if host is not None:
        host.free()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
