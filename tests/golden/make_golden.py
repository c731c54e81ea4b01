#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the authoring container only:

    python tests/golden/make_golden.py

Sources of the expected values
  ref_params.json, ref_quu.json, ref_polyfit.json, ref_gridmap.json, ref_blur.json, ref_gridmap_linear.json
      — produced by the REFERENCE's own code compiled in place (oracle/Makefile `make ref` → oracle/_ref/*.so):
        Parameters.cpp, the vendored Eigen 3.2.10 (EigenSolver, ColPivHouseholderQR) and grid_map_core.
  survey_known_answers.json
      — the three known-answer solves recorded in SURVEY.md §8(c) (outputs of the reference solver run during the
        survey); written out here verbatim as data.
  oracle_solves.json
      — outputs of oracle/ (the CPU restatement, itself pinned by the two groups above) on small seeded scenes;
        regression vectors for the HIP path when neither /root/reference nor a compiler is around.  NOT reference output.
The fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd"))

from oracle import oracle as O  # noqa: E402

dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def gen_params():
    L = O.ref_lib("params")
    p = O.Params()
    L.ref_params_default(C.byref(p))
    d = p.as_dict()
    d.pop("lamb_factor")
    d.pop("lamb_max")
    dump("ref_params.json", {"source": "I/Parameters.cpp:3-75 compiled in place", "params": d})


def gen_quu():
    L = O.ref_lib("eigen")
    rng = np.random.Generator(np.random.PCG64(11))
    cases = []
    mats = []
    for _ in range(40):  # SPD, symmetric to rounding
        a = rng.uniform(2, 50)
        d = rng.uniform(8, 80)
        b = rng.uniform(-1, 1) * np.sqrt(a * d) * 0.9
        mats.append([a, b, b, d])
    for _ in range(20):  # SPD with rounding-level asymmetry, as (fu*V)*fu' produces
        a = rng.uniform(2, 50)
        d = rng.uniform(8, 80)
        b = rng.uniform(-1, 1) * np.sqrt(a * d) * 0.9
        mats.append([a, b * (1 + 2e-16), b, d])
    for _ in range(20):  # indefinite: one eigenvalue clamped at 0
        a = rng.uniform(-5, 5)
        d = rng.uniform(-5, 5)
        b = rng.uniform(-6, 6)
        mats.append([a, b, b, d])
    mats += [[2.0, 0.0, 0.0, 8.0], [8.0, 0.0, 0.0, 2.0], [3.0, 1e-17, 1e-17, 3.0], [5.0, 1e-20, 0.0, 9.0],
             [2.0, 0.0, 1e-3, 8.0], [4.0, 2.0, 2.0, 4.0], [1e-9, 0.0, 0.0, 1e9]]
    for m in mats:
        for lamb in (1.0, 1e-3, 1e4):
            Q = np.array(m, dtype=np.float64)  # column-major: [q00, q10, q01, q11]
            Qinv = np.zeros(4)
            ev = np.zeros(2)
            vec = np.zeros(4)
            rc = L.ref_quu_inverse(Q.ctypes.data_as(dp), C.c_double(lamb), Qinv.ctypes.data_as(dp), ev.ctypes.data_as(dp),
                                   vec.ctypes.data_as(dp))
            cases.append(dict(Quu=Q.tolist(), lamb=lamb, rc=rc, Qinv=Qinv.tolist(), eval=ev.tolist(), evec=vec.tolist()))
    dump("ref_quu.json", {"source": "I/iLQR.cpp:155-175 over the reference's vendored Eigen 3.2.10 EigenSolver",
                          "layout": "column-major 2x2", "cases": cases})


def gen_polyfit():
    L = O.ref_lib("eigen")
    rng = np.random.Generator(np.random.PCG64(12))
    cases = []

    def add(x, y, deg=5):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        c = np.zeros(deg + 1)
        L.ref_polyfit(x.ctypes.data_as(dp), y.ctypes.data_as(dp), int(x.size), int(deg), c.ctypes.data_as(dp))
        cases.append(dict(x=x.tolist(), y=y.tolist(), degree=deg, coeffs=c.tolist()))

    x = np.arange(20.0)
    add(x, 0.5 * np.sin(0.05 * x))                      # the known-answer scene's first tick
    for _ in range(6):                                   # C2-style paths at the origin
        A, om, ph = rng.uniform(0, 1.5), rng.uniform(0.02, 0.08), rng.uniform(0, 2 * np.pi)
        add(x, A * np.sin(om * x + ph))
    for x0 in (57.0, 120.0, 180.0):                      # later ticks: global-frame x (ill-conditioned Vandermonde)
        xs = x0 + np.arange(20.0)
        add(xs, 0.5 * np.sin(0.05 * xs))
    xs = 185.0 + np.arange(15.0)                         # end of path: fewer than 20 waypoints
    add(xs, 0.5 * np.sin(0.05 * xs))
    add(np.arange(4.0), np.array([0.0, 1.0, 0.5, 2.0]))  # fewer rows than columns
    add(x * 0.37 - 3.0, rng.normal(size=20))             # noisy data
    dump("ref_polyfit.json", {"source": "I/LocalPlanner.cpp:101-117 over the reference's vendored Eigen 3.2.10 "
                                        "ColPivHouseholderQR", "cases": cases})


def gen_gridmap():
    L = O.ref_lib("gridmap")
    rng = np.random.Generator(np.random.PCG64(13))
    out = {"source": "G/grid_map_core compiled in place; warp recipe M/src/local_costmap.cpp:242-264",
           "geometry": [], "position": [], "index": [], "warp": []}
    geoms = [(8.0, 6.0, 1.0, 1.0, -2.0), (3.0, 2.0, 1.0, 1.5, 0.0), (30.0, 20.0, 0.2, 15.0, 0.0),
             (301.2, 301.2, 0.2, 93.14, -205.96), (5.05, 2.95, 0.1, -0.3, 0.7), (3.0, 2.0, 1.0, -12.4, -7.1)]
    for g in geoms:
        r, c = C.c_int(), C.c_int()
        lx, ly = C.c_double(), C.c_double()
        L.ref_geometry(*map(C.c_double, g), C.byref(r), C.byref(c), C.byref(lx), C.byref(ly))
        out["geometry"].append(dict(args=list(g), rows=r.value, cols=c.value, len_x=lx.value, len_y=ly.value))
        for _ in range(12):
            i, j = int(rng.integers(-1, r.value + 1)), int(rng.integers(-1, c.value + 1))
            px, py = C.c_double(), C.c_double()
            ok = L.ref_get_position(*map(C.c_double, g), i, j, C.byref(px), C.byref(py))
            out["position"].append(dict(geom=list(g), i=i, j=j, ok=ok, x=px.value if ok else None, y=py.value if ok else None))
        for _ in range(40):
            qx = g[3] + rng.uniform(-0.6, 0.6) * g[0]
            qy = g[4] + rng.uniform(-0.6, 0.6) * g[1]
            if rng.random() < 0.3:  # exactly on cell boundaries
                qx = g[3] + round((qx - g[3]) / g[2]) * g[2]
                qy = g[4] + round((qy - g[4]) / g[2]) * g[2]
            ii, jj = C.c_int(), C.c_int()
            ok = L.ref_get_index(*map(C.c_double, g), C.c_double(qx), C.c_double(qy), C.byref(ii), C.byref(jj))
            out["index"].append(dict(geom=list(g), x=qx, y=qy, ok=ok, i=ii.value, j=jj.value))

    def warp_case(sg, dg, pose, src, bbox=None):
        sr, sc, dr, dc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        t0, t1 = C.c_double(), C.c_double()
        L.ref_geometry(*map(C.c_double, sg), C.byref(sr), C.byref(sc), C.byref(t0), C.byref(t1))
        L.ref_geometry(*map(C.c_double, dg), C.byref(dr), C.byref(dc), C.byref(t0), C.byref(t1))
        src = np.asfortranarray(src(sr.value, sc.value), dtype=np.float32)
        dst = np.zeros((dr.value, dc.value), dtype=np.float32, order="F")
        bb = None
        if bbox is not None:
            bbox = np.asfortranarray(bbox(dr.value, dc.value), dtype=np.float32)
            bb = bbox.ctypes.data_as(fp)
        oob = L.ref_warp(src.ctypes.data_as(fp), *map(C.c_double, sg), dst.ctypes.data_as(fp), *map(C.c_double, dg),
                         *map(C.c_double, pose), bb)

        def enc(a):
            return [None if not np.isfinite(v) else float(v) for v in a.flatten(order="F")]
        out["warp"].append(dict(src_geom=list(sg), dst_geom=list(dg), pose=list(pose), src_shape=list(src.shape),
                                dst_shape=list(dst.shape), src=enc(src), bbox=None if bbox is None else enc(bbox),
                                dst=enc(dst), n_out_of_range=int(oob)))

    ramp = lambda r, c: np.add.outer(10.0 * np.arange(r), np.arange(c))  # noqa: E731
    warp_case((8.0, 6.0, 1.0, 1.0, -2.0), (3.0, 2.0, 1.0, 1.5, 0.0), (0.5, -2.2, 0.3), ramp)  # SURVEY §8(c) known answer
    rnd = lambda r, c: rng.integers(0, 101, (r, c)).astype(np.float32)  # noqa: E731
    for k in range(6):
        th = float(rng.uniform(-np.pi, np.pi))
        warp_case((20.0, 16.0, 0.5, 2.0, -1.0), (6.0, 4.0, 0.25, 3.0, 0.0),
                  (2.0 + float(rng.uniform(-3, 3)), -1.0 + float(rng.uniform(-3, 3)), th), rnd)
    # partly outside the source, with NaN payload and a bbox layer
    nanmap = lambda r, c: np.where(rng.random((r, c)) < 0.1, np.nan, rng.integers(0, 101, (r, c))).astype(np.float32)  # noqa: E731
    box = lambda r, c: np.where(rng.random((r, c)) < 0.2, 100.0, 0.0).astype(np.float32)  # noqa: E731
    warp_case((10.0, 10.0, 0.5, 0.0, 0.0), (8.0, 6.0, 0.5, 4.0, 0.0), (3.0, 3.0, 0.7), nanmap, box)
    warp_case((10.0, 10.0, 0.5, 0.0, 0.0), (4.0, 4.0, 0.5, 0.0, 0.0), (0.0, 0.0, np.pi / 2), rnd)  # cell-boundary hits
    warp_case((10.0, 10.0, 0.5, 0.0, 0.0), (4.0, 4.0, 0.5, 0.0, 0.0), (100.0, 100.0, 0.0), rnd)  # fully outside
    dump("ref_gridmap.json", out)


def gen_survey():
    dump("survey_known_answers.json", {
        "source": "SURVEY.md §8(c): outputs of the reference solver run during the survey",
        "scene": "Parameters defaults with horizon=N; path (i, 0.5 sin(0.05 i)), i=0..199; x0=(0,0.1,3.0,0.02); "
                 "obstacle o: dim (4.79,2.16), pose (15+12o, o odd ? -1.0 : 0.8, 0, 0.1 o) for all t; default control_seq",
        "cases": [
            dict(N=30, M=2, iterations=15, exit="lambda_max", U0=[2.3192183472726575, -0.081980325161252365],
                 XN=[12.430960211, 3.388616849, 4.805863543, 0.482461480]),
            dict(N=50, M=4, iterations=9, exit="lambda_max", U0=[1.5637306265576001, -0.09511613452517606],
                 XN=[21.388494959, -0.296203534, 4.937663127, 0.326099768]),
            dict(N=80, M=16, iterations=20, exit="max_iter", U0=[1.7758881112037488, 0.2133815668197884],
                 XN=[11.978169525, 4.149888973, 4.820220232, 3.173218090]),
            dict(N=50, M=0, iterations=15, exit=None, U0=None, XN=None),
        ]})


def gen_oracle_solves():
    from cilqr_amd import scenes
    out = {"source": "oracle/ (CPU restatement) — regression vectors, NOT reference output", "cases": []}
    for name, N, M, B, seed in (("c2", 50, 4, 8, scenes.SEED0 + 2), ("n30", 30, 2, 4, 77), ("c5", 80, 16, 2, scenes.SEED0 + 5),
                                ("m0", 50, 0, 4, 78)):
        p = O.default_params(N)
        sc = scenes.make_static(B, N, M, p, seed, local_plan=O.local_plan)
        r = O.solve_batch(p, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, threads=4)
        out["cases"].append(dict(name=name, N=N, M=M, B=B, seed=seed,
                                 inputs={k: (None if sc[k] is None else np.asarray(sc[k]).reshape(B, -1).tolist())
                                         for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")},
                                 U=r["U"].tolist(), X=r["X"].tolist(), J=r["J"].tolist(), iters=r["iters"].tolist(),
                                 status=r["status"].tolist()))
    dump("oracle_solves.json", out)


def gen_blur():
    """Small maps through the reference's grid_map_core EllipseIterator + vendored Eigen float EigenSolver."""
    rng = np.random.Generator(np.random.PCG64(14))
    out = {"source": "M/src/arbitrary_transformation.cu:8-157 recipe over the reference's grid_map_core + Eigen 3.2.10 (oracle/_ref)",
           "note": "cells listed in `edge_ub` have a bounding box clamped onto the far map edge: the reference then walks the "
                   "non-existent row/column `size` through an uninitialised Position (undefined behaviour) — excluded from parity",
           "cases": []}
    for geom, sig, th, index in (((6.0, 4.0, 0.2, 3.0, 0.0), (0.16, 0.16, 0.017), 0.3, 0),
                                 ((6.0, 4.0, 0.2, 3.0, 0.0), (0.005, 0.005, 0.0125), -1.2, 7),
                                 ((5.0, 3.0, 0.1, 1.0, -0.5), (0.05, 0.08, 0.03), 2.5, 0),
                                 ((6.0, 4.0, 0.2, 2.0, 0.0), (0.3, 0.2, 0.05), 0.9, 0)):
        g = O.map_geom(*geom)
        src = rng.integers(0, 101, (g.rows, g.cols)).astype(np.float32)
        src[rng.random(src.shape) < 0.02] = np.nan
        ro, rc, ell = O.ref_blur(src, geom, np.sin(th), np.cos(th), *sig, index=index)
        oo, oc, _ = O.blur(src, g, np.sin(th), np.cos(th), *sig, index=index)
        edge = np.nonzero(rc != oc)[0]

        def enc(a):
            return [None if not np.isfinite(v) else float(v) for v in np.asarray(a).flatten(order="F")]
        out["cases"].append(dict(geom=list(geom), sigma=list(sig), theta=th, index=index, shape=[g.rows, g.cols], src=enc(src),
                                 out=enc(ro), count=rc.tolist(), ellipse=[[None if not np.isfinite(v) else float(v) for v in e] for e in ell],
                                 edge_ub=edge.tolist()))
    dump("ref_blur.json", out)


def gen_gridmap_linear():
    """GridMap::atPosition(..., INTER_LINEAR) of the reference's grid_map_core at random positions: pins the lookup the
    uncertainty cost is built on (the cost's own arithmetic has no reference source: parity unpinned)."""
    rng = np.random.Generator(np.random.PCG64(15))
    out = {"source": "G/grid_map_core/src/GridMap.cpp:191-201,770-837 (atPosition with INTER_LINEAR) compiled in place (oracle/_ref)",
           "cases": []}
    for geom in ((6.0, 4.0, 0.2, 3.0, 0.0), (5.0, 3.0, 0.1, 1.0, -0.5)):
        g = O.map_geom(*geom)
        src = (rng.random((g.rows, g.cols)) * 100).astype(np.float32)
        n = 150
        # interior positions: at least one cell away from the border (the library's own edge handling differs)
        qx = geom[3] + (rng.random(n) - 0.5) * (g.len_x - 4 * g.res)
        qy = geom[4] + (rng.random(n) - 0.5) * (g.len_y - 4 * g.res)
        val, ok = O.ref_linear(src, geom, qx, qy)
        assert ok.all()
        out["cases"].append(dict(geom=list(geom), shape=[g.rows, g.cols], src=[float(v) for v in src.flatten(order="F")],
                                 qx=qx.tolist(), qy=qy.tolist(), value=[float(v) for v in val]))
    dump("ref_gridmap_linear.json", out)


if __name__ == "__main__":
    O.build(ref=True)
    gen_survey()
    gen_params()
    gen_quu()
    gen_polyfit()
    gen_gridmap()
    gen_blur()
    gen_gridmap_linear()
    gen_oracle_solves()
