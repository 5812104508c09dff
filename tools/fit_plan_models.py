"""Reproduces the constants of the fitted dispatch models from the committed calibration data (no GPU):

    python tools/fit_plan_models.py nv [profiles/calib_nv_small_r3.txt]          -> gemm_nvf4.hip.h: nvf4_plan (tile kernels A / B / E, reduce pass)
    python tools/fit_plan_models.py nv-skinny [profiles/calib_nv_small_r3_graph.txt] -> the small-batch kernel's price (GPU-only re-take)
    python tools/fit_plan_models.py mx [profiles/calib_mx_small_r3.txt]          -> capi.hip: plan_small (ring kernels, per format)

Model of a tile kernel:  t = a_c + g b_c kt / 16  [+ r0 + (S + 1) M N 4 bytes / bw  for the reduce pass of S > 1 K ranges]
with kt = K stages per workgroup and g = 1 while workgroups <= CUs, else ceil(workgroups / CUs) e_c.  Least squares on log(model / measured), soft-L1 loss.
The functions are importable (tests/test_cabi_and_host.py checks that the constants in the C++ sources are the ones this script finds)."""
import math
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS = 256


def load(path, formats=False):
    rows, names = [], None
    for line in open(path):
        if line.startswith("#"):
            names = line.split("|")[1].split()
            continue
        head, vals = line.split("|")[:2]
        h = head.split()
        d = dict(zip(names, (float(x) for x in vals.split())))
        rows.append((h[0], int(h[1]), int(h[2]), int(h[3]), d))
    return names, rows


def _tile_model(th, ntile, tile_index, bm, bn, m, n, kt_total, S, even):
    a, b, e = th[tile_index], th[ntile + tile_index], th[2 * ntile + tile_index]
    r0, bw = th[3 * ntile], th[3 * ntile + 1]
    if even:   # NVFP4: ranges of an even number of stages
        per = 2 * math.ceil(kt_total / (2 * S))
    else:
        per = math.ceil(kt_total / S)
    s2 = math.ceil(kt_total / per)
    nn = math.ceil(m / bm) * math.ceil(n / bn) * s2 / CUS
    g = 1.0 if nn <= 1 else math.ceil(nn) * e
    t = a + g * b * per / 16
    if s2 > 1:
        t += r0 + (s2 + 1) * m * n * 4 / (bw * 1e6)
    return max(t, 0.1)


def fit_nv(path=None):
    """-> (a[3], b[3], e[3], r0, bw) for 128x128, 128x64, 64x64 tiles."""
    names, rows = load(path or os.path.join(ROOT, "profiles", "calib_nv_small_r3.txt"))
    tiles = {"128x128": (128, 128, 0), "128x64": (128, 64, 1), "64x64": (64, 64, 2)}
    cands = []
    for nm in names:
        base, _, s = nm.partition("/")
        if base in tiles:
            cands.append((nm, tiles[base], int(s) if s else 1))

    def resid(th):
        r = []
        for _, m, n, k, d in rows:
            for nm, (bm, bn, ci), S in cands:
                if not math.isnan(d[nm]):
                    r.append(math.log(_tile_model(th, 3, ci, bm, bn, m, n, math.ceil(k / 256), S, True) / d[nm]))
        return np.array(r)

    x0 = np.array([3.1, 2.5, 2.2, 29.7, 20.9, 14.9, 1.0, 0.9, 0.8, 5.0, 3.5])
    res = least_squares(resid, x0, loss="soft_l1", f_scale=0.1)
    th = res.x
    return th[0:3], th[3:6], th[6:9], th[9], th[10], float(np.sqrt((resid(th) ** 2).mean()))


def fit_nv_skinny(path=None):
    """-> (s0, s1): t = s0 + s1 ceil(32x32 workgroups / CUs) K / 4096."""
    _, rows = load(path or os.path.join(ROOT, "profiles", "calib_nv_small_r3_graph.txt"))
    data = [(m, n, k, d["skinny"]) for _, m, n, k, d in rows if not math.isnan(d["skinny"])]
    f = lambda t, m, n, k: t[0] + t[1] * math.ceil(math.ceil(m / 32) * math.ceil(n / 32) / CUS) * k / 4096
    res = least_squares(lambda t: np.array([math.log(f(t, m, n, k) / x) for m, n, k, x in data]), np.array([3.0, 4.4]), loss="soft_l1", f_scale=0.1)
    return res.x[0], res.x[1], float(np.sqrt((res.fun ** 2).mean()))


def fit_mx(path=None):
    """-> {fmt: (a[3], b[3], e[3], r0, bw, rms)} for 64x64, 64x128, 128x128 ring tiles; measurements >= 13 us only (below: the calibration's Python caller)."""
    names, rows = load(path or os.path.join(ROOT, "profiles", "calib_mx_small_r3.txt"))
    tiles = {"r64": (64, 64, 0), "r64x128": (64, 128, 1), "r128": (128, 128, 2)}
    cands = []
    for nm in names:
        base, _, s = nm.partition("/")
        if base in tiles:
            cands.append((nm, tiles[base], int(s) if s else 1))
    out = {}
    for fmt, ebits in (("mxf4", 4), ("mxf8", 8)):
        R = [r for r in rows if r[0] == fmt]

        def resid(th):
            r = []
            for _, m, n, k, d in R:
                for nm, (bm, bn, ci), S in cands:
                    x = d[nm]
                    if not math.isnan(x) and x >= 13.0:
                        r.append(math.log(_tile_model(th, 3, ci, bm, bn, m, n, math.ceil(k * ebits / 8 / 128), S, False) / x))
            return np.array(r)

        x0 = np.array([3, 3, 3, 5, 8, 10, 0.8, 0.8, 0.9, 4.0, 4.0], dtype=float)
        res = least_squares(resid, x0, loss="soft_l1", f_scale=0.1, bounds=([0, 0, 0, 0.5, 0.5, 0.5, 0.3, 0.3, 0.3, 0, 0.5], [20, 20, 20, 60, 60, 60, 1.2, 1.2, 1.2, 20, 20]))
        th = res.x
        out[fmt] = (th[0:3], th[3:6], th[6:9], th[9], th[10], float(np.sqrt((resid(th) ** 2).mean())))
    return out


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "nv"
    path = sys.argv[2] if len(sys.argv) > 2 else None
    r3 = lambda v: [round(float(x), 3) for x in v]
    if what == "nv":
        a, b, e, r0, bw, rms = fit_nv(path)
        print("nvf4_plan  128x128 / 128x64 / 64x64:  A", r3(a), " B", r3(b), " E", r3(e), " reduce %.3f us + bytes / %.3f TB/s   rms %.3f" % (r0, bw, rms))
    elif what == "nv-skinny":
        s0, s1, rms = fit_nv_skinny(path)
        print("nvf4_plan  skinny: %.3f + %.3f ceil(workgroups / CUs) K / 4096   rms %.3f" % (s0, s1, rms))
    elif what == "mx":
        for fmt, (a, b, e, r0, bw, rms) in fit_mx(path).items():
            print("plan_small %s  64x64 / 64x128 / 128x128:  A" % fmt, r3(a), " B", r3(b), " E", r3(e), " reduce %.3f us + bytes / %.3f TB/s   rms %.3f" % (r0, bw, rms))
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
