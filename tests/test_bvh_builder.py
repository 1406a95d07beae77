"""Host-side SAH builder of the fast mode (raytracing_amd/csrc/rt_bvh_build.h; no reference counterpart -- the reference's builder,
bvh.hpp:22-56, stays resident for the parity modes and is checked in test_host_mirror.py).  The header is plain C++: a small
harness (tests/bvh_builder_check.cpp) is compiled with g++ and walks the finished tree."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "raytracing_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_builder_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(HERE, "bvh_builder_check.cpp")])
    return exe


def run(exe, n, limit, bins, leaf_max, budget=35):
    out = subprocess.check_output([exe, str(n), str(limit), str(bins), str(leaf_max), str(budget)], text=True)
    return {k: int(v) for k, v in (kv.split("=") for kv in out.split())}


@pytest.mark.parametrize("n,limit,bins,leaf_max", [(1, 26, 16, 4), (2, 26, 16, 4), (5, 26, 64, 4), (1000, 26, 16, 4), (200000, 26, 64, 4),
                                                   (200000, 18, 64, 4), (200000, 17, 16, 8), (50000, 40, 16, 1)])
def test_tree_is_a_partition_with_containing_boxes_within_the_depth_limit(harness, n, limit, bins, leaf_max):
    """Every primitive sits in exactly one leaf, every node is reached once, every box a parent stores contains what hangs below
    it, leaves hold at most leaf_max records, the depth the builder reports is the depth of the deepest leaf, and it respects the
    limit whenever halving could (1 + ceil(log2(n / leaf_max)) levels) -- down to the tightest limit that allows."""
    r = run(harness, n, limit, bins, leaf_max)
    assert r["missing"] == 0 and r["repeated"] == (1 if n == 1 else 0)       # a single primitive hangs under both children of its root
    assert r["unreachable"] == 0 and r["shared"] == 0 and r["bad_box"] == 0 and r["bad_leaf"] == 0
    if n > leaf_max:
        assert r["depth"] == r["walked_depth"]          # root at level 1, a leaf one below the node that names it
    else:
        assert r["walked_depth"] - 1 <= r["depth"] <= r["walked_depth"]     # the one-node wrappers report the root's level
    assert r["depth"] <= limit, r


@pytest.mark.parametrize("n,limit,budget", [(1, 26, 35), (2, 26, 35), (3, 26, 35), (5, 26, 35), (1000, 26, 35), (200000, 26, 35), (200000, 26, 26),
                                            (200000, 18, 17), (200000, 26, 5), (50000, 40, 35)])
def test_wide_tree_is_the_same_partition_within_its_stack_budget(harness, n, limit, budget):
    """rtb::widen, the collapse to four-wide nodes: every primitive still sits in exactly one leaf, every wide node is reached once,
    every box a node stores contains what hangs below it, nodes hold 2..4 children, and no traversal -- whatever order it visits
    children in, with every unvisited sibling waiting on the stack -- holds more entries than the tree reports.  The report respects
    the budget wherever the binary tree does (a budget below the binary tree's own need leaves the tallest paths binary)."""
    r = run(harness, n, limit, 64, 4, budget)
    assert r["w_missing"] == 0 and r["w_repeated"] == 0 and r["w_unreachable"] == 0 and r["w_shared"] == 0
    assert r["w_bad_box"] == 0 and r["w_bad_n"] == 0
    assert r["w_unsorted"] == 0, "slot order is visiting order: children ascend along the node's sort axis (rt_wave.h node_step compares no distances)"
    assert r["w_worst_stack"] <= r["w_stack"]
    assert r["w_stack"] <= max(budget, r["depth"] - 1), r
    if n >= 1000 and budget >= 35:
        assert r["w_nodes"] * 3 < r["nodes"] * 2 and r["w_full"] * 3 > r["w_nodes"], r          # it does collapse (single-triangle leaves cannot be opened)


def test_impossible_limit_falls_back_to_the_stack_maximum(harness):
    r = run(harness, 200000, 10, 16, 4)        # 2^10 leaves of 4 cannot hold 200 000 primitives
    assert r["missing"] == 0 and r["repeated"] == 0 and r["bad_box"] == 0
    assert 10 < r["depth"] <= 42


def test_quantised_axis_contains_its_boxes(tmp_path):
    """rtb::quantise_axis_wide, the encoder of the quantised four-wide nodes: over magnitudes 0 ... 1e12, box sizes 0 ... 1e4, both
    signs, one to four children, boxes that straddle zero and degenerate ones, the planes the kernel decodes (whole dword as the
    origin -- exponent byte included -- plus a byte offset times 2^e) contain every child's planes, lo <= hi byte-wise (what lets a
    ray pick the near plane by the sign of its direction alone), no plane is clipped at 255, none is looser than one step beyond
    what the origin's low byte gives away, and unused slots hold the inverted box."""
    exe = str(tmp_path / "qnode_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(HERE, "qnode_check.cpp")])
    r = {k: int(v) for k, v in (kv.split("=") for kv in subprocess.check_output([exe], text=True).split())}
    assert r["cases"] > 30000
    assert r["not_contained"] == 0 and r["disordered"] == 0 and r["clipped"] == 0 and r["loose"] == 0 and r["bad_empty"] == 0, r


def test_visit_order_census_tells_a_surface_from_a_soup(tmp_path):
    """rtb::visit_order_census (what rtcore.hip asks before it picks the wide nodes' visiting order): on a tessellated surface the slot
    order costs incoherent rays about as many box tests as the distance order (< 5 % more: slot order, no sort in the kernel), in a sparse
    soup of small triangles clearly more (> 5 %: the sorted visit)."""
    exe = str(tmp_path / "visit_order_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(HERE, "visit_order_check.cpp")])
    r = {k: float(v) for k, v in (kv.split("=") for kv in subprocess.check_output([exe], text=True).split())}
    assert r["mesh_dist"] <= r["mesh_slot"] <= 1.05 * r["mesh_dist"], r
    assert r["soup_slot"] > 1.05 * r["soup_dist"], r
