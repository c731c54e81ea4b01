#!/usr/bin/env python3
"""Grouped family: kernel time for every lane grouping G at large batches (CILQR_FORCE_G), beside the library's own pick.
The rule in cilqr_api.cpp (pick_group_lanes) is drawn from this table.   python tools/group_lanes_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

def run(sc, N, M, B, force):
    if force: os.environ["CILQR_FORCE_G"] = str(force)
    else: os.environ.pop("CILQR_FORCE_G", None)
    p = cilqr_amd.default_params(N)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x0, U0, poly, xpl, pose, dim = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"]), dv(sc["obs_pose"]), dv(sc["obs_dim"])
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U = U0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        U.copy_(U0); torch.cuda.synchronize(); e0.record()
        s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                             pose.data_ptr(), dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    g = s.solve_family(B, N, M)
    s.close()
    return best, g

shapes = [(50, 4), (80, 16), (30, 2), (80, 4)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in q.split("x")) for q in os.environ["SHAPES"].split(",")]
for N, M in shapes:
    for B in (8192, 16384, 32768, 65536):
        p = cilqr_amd.default_params(N)
        sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
        _, pick = run(sc, N, M, B, 0)
        row = []
        for G in (1, 2, 4, 8, 16, 32, 64):
            if G * B < 65536:  # fewer wavefronts than SIMDs
                continue
            t, _ = run(sc, N, M, B, G)
            row.append((G, t))
        best = min(t for _, t in row)
        print("N=%3d M=%2d B=%5d: " % (N, M, B) + " | ".join("G=%2d %7.3f ms%s" % (G, t, " <" if t == best else "") for G, t in row)
              + " || rule picks G=%d: loses %.1f %%" % (pick, 100 * (dict(row)[pick] / best - 1)), flush=True)
