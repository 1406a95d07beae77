"""GPU test of rt_first_hit_noalloc (include/rtcore.h): the variant of rt_first_hit whose device-side staging lives in a static
device array of the library instead of hipMalloc'd memory.  Through the C ABI it must give rt_first_hit's outputs bit for bit, on
two calls in a row, the second with a smaller input (the array keeps the first call's contents), and it must refuse a batch beyond
its stated maximum.  Named to be collected last: the round's end runs pytest -x."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

KEYS = ("prim", "front", "nbox", "nleaf", "hash")


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in KEYS) and np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64))


def test_first_hit_noalloc_matches_first_hit(golden_dir):
    import raytracing_amd as R
    from raytracing_amd.rtcore import NOALLOC_MAX_RAYS
    R.load_library()
    g = np.load(os.path.join(golden_dir, "c0_seed1.npz"))
    m = {k: int(g["meta_" + k]) for k in ("width", "height", "seed")}
    osc = O.Scene.c0(m["width"], m["height"], m["seed"])
    sc = R.Scene(osc.export())
    rays = g["c0_fh_rays"].reshape(-1, 7)
    assert 1000 < len(rays) <= NOALLOC_MAX_RAYS
    want_all = sc.first_hit(rays)
    got_all = sc.first_hit(rays, noalloc=True)                     # first call: the whole fixture
    assert same(got_all, want_all)
    assert np.array_equal(got_all["t"].view(np.uint64), g["c0_fh_t"].view(np.uint64)), "and both equal the reference fixture"
    small = rays[37:37 + 301][::-1].copy()                          # second call: fewer, different rays over the same array
    assert same(sc.first_hit(small, noalloc=True), sc.first_hit(small))
    assert same(sc.first_hit(rays[:1], noalloc=True), sc.first_hit(rays[:1]))
    # a full-size batch, then the fixture again
    rng = np.random.default_rng(5)
    big = rays[rng.integers(0, len(rays), NOALLOC_MAX_RAYS)]
    assert same(sc.first_hit(big, noalloc=True), sc.first_hit(big))
    assert same(sc.first_hit(rays, noalloc=True), want_all)
    with pytest.raises(R.RtError, match="RT_NOALLOC_MAX_RAYS"):
        sc.first_hit(np.concatenate([big, rays[:1]]), noalloc=True)
    assert same(sc.first_hit(rays[:0], noalloc=True), sc.first_hit(rays[:0]))     # empty batch: no launch, no error
    sc.destroy()
    osc.free()
