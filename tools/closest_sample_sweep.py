#!/usr/bin/env python3
"""The closest-sample search against the full scan (cilqr_debug_closest_sample) on many more random queries than the test suite
runs: SEEDS × 1 M queries of the test's mixture, plus extreme families (tiny and huge sample spacing, points 1e3 m away, nearly
circular paths around the point).  Prints disagreements (there must be none) and the share decided by Newton.

    python tools/closest_sample_sweep.py [seeds]
"""
import os, sys
from math import comb
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np
import cilqr_amd

SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = cilqr_amd.default_params(50)
s = cilqr_amd.Solver(p, max_batch=64, max_horizon=50, max_obstacles=1)
total = bad_total = 0
for seed in range(SEEDS):
    rng = np.random.default_rng(77000 + seed)
    n = 1_000_000
    q = np.zeros((n, 10))
    xf = rng.uniform(-50.0, 50.0, n)
    fam = rng.integers(0, 6, n)
    length = rng.uniform(8.0, 30.0, n)
    length = np.where(fam == 4, rng.uniform(1e-3, 0.5, n), length)     # tiny spacing
    length = np.where(fam == 5, rng.uniform(200.0, 5000.0, n), length)  # huge spacing
    length *= np.where(rng.random(n) < 0.1, -1.0, 1.0)
    slope = rng.uniform(-2.0, 2.0, n) * (fam != 0) + rng.uniform(-0.3, 0.3, n) * (fam == 0)
    curv = rng.uniform(-0.2, 0.2, n) * ((fam == 2) | (fam == 3)) + rng.uniform(-0.01, 0.01, n) * (fam == 0)
    c3 = rng.uniform(-0.004, 0.004, n) * (fam == 3)
    c4 = rng.uniform(-2e-4, 2e-4, n) * (fam == 3)
    c5 = rng.uniform(-2e-6, 2e-6, n) * (fam == 3)
    y0 = rng.uniform(-3.0, 3.0, n)
    cu = np.stack([y0, slope, curv, c3, c4, c5], axis=1)
    for j in range(6):
        for i in range(j + 1):
            q[:, i] += cu[:, j] * comb(j, i) * (-xf) ** (j - i)
    q[:, 6] = xf
    q[:, 7] = xf + length
    along = rng.uniform(-0.3, 1.3, n) * length
    yp = sum(cu[:, j] * along ** j for j in range(6))
    r = rng.random(n)
    lateral = np.where(r < 0.15, 0.0, np.where(r < 0.9, rng.uniform(-10.0, 10.0, n), rng.uniform(-1000.0, 1000.0, n)))
    # the point at the centre of curvature of the path's start (every sample about equally far: the worst case for any pruning)
    cc = (fam == 2) & (rng.random(n) < 0.3) & (np.abs(curv) > 1e-3)
    lateral = np.where(cc, 1.0 / (2.0 * curv + 1e-300) * (1.0 + slope ** 2) ** 1.5 / np.sqrt(1.0 + slope ** 2), lateral)
    q[:, 8] = xf + along * np.where(cc, 0.0, 1.0) - np.where(cc, lateral * slope / np.sqrt(1 + slope ** 2), 0.0)
    q[:, 9] = yp * np.where(cc, 0.0, 1.0) + np.where(cc, y0, 0.0) + lateral / np.where(cc, np.sqrt(1 + slope ** 2), 1.0)
    out = s.debug_closest_sample(q)
    bad = np.nonzero(out[:, 0] != out[:, 1])[0]
    total += n
    bad_total += bad.size
    print("seed %d: %d queries, %d disagreements, Newton decided %.1f %%; by family: %s" % (
        seed, n, bad.size, 100 * out[:, 2].mean(), " ".join("%d:%.0f%%" % (f, 100 * out[fam == f, 2].mean()) for f in range(6))), flush=True)
    for b in bad[:3]:
        print("   ", repr(q[b].tolist()), out[b])
print("total %d queries, %d disagreements" % (total, bad_total))
s.close()
sys.exit(1 if bad_total else 0)
