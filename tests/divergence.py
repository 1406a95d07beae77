"""Where a GPU fp64 path parts from the CPU oracle's path on the SAME random stream, and why (tools/first_divergence.py, test_gpu_parity.py).

Both sides can list the rays of one (pixel, sample) path: the oracle (orc_debug_path) and the HIP core (rt_debug_path: the reference-order walk with the
render kernel's own functions).  + - * / sqrt are correctly rounded on both sides, so two paths on one stream can only part where a libm call
(sin, cos, acos, atan2, pow, log) returned different last bits on the host (glibc) and on the device (ocml): the oracle lists its libm calls per
segment (orc_debug_libm), the device evaluates the same arguments (rt_debug_libm), and the calls whose results differ are the explanation.
A path that parts WITHOUT such a call would be a real defect; `explain_pixel` reports it as unexplained."""
import numpy as np

OPS = {1: "sin", 2: "cos", 3: "acos", 4: "atan2", 5: "pow", 6: "log"}


def ulps(a, b):
    """distance of two doubles in units in the last place (bit patterns as ordered integers); NaN vs NaN = 0"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ia = a.view(np.int64).copy(); ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-2 ** 63) - ia, ia); ib = np.where(ib < 0, np.int64(-2 ** 63) - ib, ib)
    d = np.abs(ia.astype(np.float64) - ib.astype(np.float64))
    return np.where(np.isnan(a) & np.isnan(b), 0.0, d)


def differing_calls(R, rows):
    """rows of the oracle's libm log (segment, op, x, y, result) -> the calls whose device result has other bits: list of dicts"""
    if len(rows) == 0:
        return []
    gpu = R.debug_libm(rows[:, 1].astype(np.int32), rows[:, 2], rows[:, 3])
    out = []
    for r, g in zip(rows, gpu):
        if np.float64(r[4]).view(np.uint64) != np.float64(g).view(np.uint64) and not (np.isnan(r[4]) and np.isnan(g)):
            out.append(dict(segment=int(r[0]), call=OPS[int(r[1])], x=float(r[2]), y=float(r[3]), cpu=float(r[4]), gpu=float(g), ulps=float(ulps(r[4], g))))
    return out


def explain_pixel(R, osc, sc, width, height, i, j, spp, seed, depth, flags=0):
    """The first sample of pixel (row i, column j) whose GPU path is not the oracle's, the first ray that differs, and the libm calls made while the
    segment before it was shaded whose results differ between host and device.  Returns a dict:
      kind      'path' (rays differ from `segment` on), 'value' (all rays identical: only a shading value can differ) or 'same' (no libm call differs either)
      sample, segment, calls (see differing_calls), explained (bool: at least one differing call, all within 2 ulp)"""
    value_calls = []
    for k in range(spp):
        cpu = osc.debug_path(i, j, k, seed, depth)
        gpu = sc.debug_path(width, height, i, j, k, seed, depth, flags)
        n = min(len(cpu), len(gpu))
        diff = [s for s in range(n) if not np.array_equal(cpu[s].view(np.uint64), gpu[s, :7].copy().view(np.uint64))]
        rows = osc.debug_libm(i, j, k, seed, depth)
        if not diff and len(cpu) == len(gpu):
            value_calls += [dict(c, sample=k) for c in differing_calls(R, rows)]
            continue
        s = diff[0] if diff else n          # ray s is the first that differs: it was made while segment s (1-based in the oracle's log) was shaded
        calls = differing_calls(R, rows[rows[:, 0] == s]) if s >= 1 else []
        return dict(kind="path", sample=k, segment=s, rays_cpu=len(cpu), rays_gpu=len(gpu), calls=calls,
                    explained=bool(calls) and max(c["ulps"] for c in calls) <= 2)
    if value_calls:
        return dict(kind="value", sample=value_calls[0]["sample"], segment=value_calls[0]["segment"], calls=value_calls, explained=max(c["ulps"] for c in value_calls) <= 2)
    return dict(kind="same", sample=-1, segment=-1, calls=[], explained=False)


def assert_same_streams(R, osc, sc, got, want, width, height, spp, seed, depth, max_frac=1e-3, flags=0, max_replayed=300):
    """The frame `got` (fp64 sums of an fp64 GPU mode) against the oracle's `want` on the same streams: all pixels within 1e-9 relative but at most
    `max_frac` of them, and every one of those (up to max_replayed, taken evenly) replayed on both sides and explained by a libm call whose
    last bits differ between host and device.  Returns the number of pixels beyond 1e-9."""
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    off = np.argwhere(rel.max(axis=2) > 1e-9)
    assert len(off) <= max_frac * width * height, f"{len(off)} of {width * height} pixels differ from the oracle beyond rounding"
    for i, j in off[::max(1, len(off) // max_replayed)]:
        e = explain_pixel(R, osc, sc, width, height, int(i), int(j), spp, seed, depth, flags)
        assert e["explained"], f"pixel ({i},{j}) differs from the oracle ({float(rel[i, j].max()):.3e}) and no libm call explains it: {e}"
    return len(off)
