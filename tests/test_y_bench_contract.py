"""bench.py itself, end to end on the GPU box at a small size: the one JSON line the driver reads, the launcher path for N > 1 (rehearsed
with gloo ranks that share the device) and the RCCL branch with a process group of one.  Kept in a file of its own that sorts after the
parity tests: these start child processes and open sockets, and the round's end runs `pytest -x`."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    import raytracing_amd
    raytracing_amd.load_library()
    return raytracing_amd


def test_bench_prints_one_json_line_with_the_contract_fields(R):
    """bench.py end to end at a small size (the generated mesh at level 3, 160 x 90, 8 spp): stdout is ONE line of JSON -- the host mirror's
    and RCCL's chatter goes to stderr -- and the line carries what the driver and the judge read: metric / value / unit, steps and
    warm-up as asked, dtype of the arithmetic, `vs_baseline` null (BASELINE.md has no published number), the workload in `config`,
    `roofline` with 0 < frac <= 1 and a kernel duration measured in this run, `cpu_baseline` with its cores, kind and sample, the image
    check of the timed scene."""
    import json
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--width", "160", "--height", "90", "--spp", "8", "--steps", "2", "--warmup", "1", "--mesh-level", "3",
           "--cpu-spp", "2", "--image-width", "80", "--image-spp", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["metric"].startswith("Mrays/s") and d["unit"] == "Mrays/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 2, 1) and d["ms_per_step"] > 0
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] in ("strong", "weak") and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["spp"] == 8
    r = d["roofline"]
    assert r["bound"] in ("valu", "hbm", "mfma") and 0 < r["frac"] <= 1 and r["achieved"] > 0 and r["peak"] > 0 and r["kernel_ms_avg"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "spp" in c["sample"]
    assert d["value"] > 3 * c["value"], "the GPU path is the one that was timed (a frame this small is launch-bound: the ratio says nothing else)"
    assert d["image"]["same_streams"]["rays_gpu"] == d["image"]["same_streams"]["rays_cpu"]
    # the headline counts rays that walked the tree; the reference's count (every trace() call, zero-direction phong rays included) rides along
    assert d["config"]["rays_per_step"] <= d["config"]["reference_rays_per_step"] and d["value"] <= d["value_reference_ray_count"]
    share = 1.0 - d["config"]["rays_per_step"] / d["config"]["reference_rays_per_step"]
    assert abs(share - d["config"]["zero_direction_ray_share"]) < 0.02, "in a scene of triangles alone every zero-direction ray is finished without a traversal"
    assert r["kernel"].startswith("rt::render_wave1_kernel<double, false, "), r["kernel"]      # 8 spp: the one-wave-workgroup twin of the triangle-scene kernel
    assert d["fp32_mode"]["dtype"] == "f32" and d["fp32_mode"]["value"] > 0


def test_bench_under_the_launcher_with_two_ranks(R):
    """bench.py as the driver starts it for N > 1 (python -m torch.distributed.run, one rank per GPU), rehearsed on this one-GPU box with two
    ranks that share the device and reduce through the host (RT_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device): rank 0
    prints the one line, n_gpus = 2, each rank rendered half of the samples, and the frame's ray count is the single-rank frame's."""
    import json
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    small = ["--width", "160", "--height", "90", "--spp", "8", "--steps", "2", "--warmup", "1", "--mesh-level", "3", "--no-cpu", "--image-width", "0", "--no-fp32"]
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[:500]
    two = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, capture_output=True, text=True, timeout=200)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and single["n_gpus"] == 1 and two["steps"] == 2 and two["value"] > 0
    assert two["config"]["rays_per_step"] == single["config"]["rays_per_step"], "streams are keyed by the global sample index: same frame for any N"
    assert "gloo" in two["config"]["parallelism"] or "2" in two["config"]["parallelism"]


def test_bench_starts_its_own_ranks_when_no_launcher_did(R):
    """Plain `python bench.py --gpus 2`, no launcher and no WORLD_SIZE: bench.py starts the two ranks itself (a child
    `python -m torch.distributed.run`, before the parent has touched the GPU), relays rank 0's one JSON line and returns the
    launcher's exit code.  Rehearsed with gloo ranks sharing the one device.  Ranks other than 0 take the scene from rank 0's
    binary cache (rt_scene_desc_save / load) instead of building it again: same frame, same ray count."""
    import json
    import sys
    small = ["--width", "160", "--height", "90", "--spp", "8", "--steps", "2", "--warmup", "1", "--mesh-level", "3", "--no-cpu", "--image-width", "0", "--no-fp32"]
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    two = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small, capture_output=True, text=True, timeout=200, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["value"] > 0
    assert two["config"]["rays_per_step"] == single["config"]["rays_per_step"] and two["config"]["reference_rays_per_step"] == single["config"]["reference_rays_per_step"]
    assert two["config"]["n_primitives"] == single["config"]["n_primitives"]
    # a launcher that fails must fail bench.py too: three ranks cannot split... they can; an unknown flag cannot be parsed by any rank
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"], capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode != 0 and not bad.stdout.strip()


def test_bench_rccl_branch_with_a_group_of_one(R):
    """What a one-GPU box can run of the RCCL branch: under the launcher with one rank and RT_BENCH_FORCE_DIST=1, bench.py builds an nccl
    (= RCCL) process group of one and issues every collective of the N > 1 path (the fp64 reduce of the frame, the barriers, the two
    all-reduces of the totals).  RCCL prints its banner on stdout; the JSON line must still be the only thing bench.py prints there."""
    import json
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    small = ["--width", "160", "--height", "90", "--spp", "8", "--steps", "2", "--warmup", "1", "--mesh-level", "3", "--no-cpu", "--image-width", "0", "--no-fp32"]
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1")
    env.pop("RT_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "nccl" in d["config"]["parallelism"].lower(), d["config"]["parallelism"]

