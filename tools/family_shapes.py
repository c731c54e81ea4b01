#!/usr/bin/env python3
"""Kernel family by shape: wall time of cilqr_solve_batch_device for static scenes of several (N, M, B), default family choice
against the one-wavefront-per-solve family forced (CILQR_FORCE_G=64).  Diagnostic tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

def run(N, M, B, force):
    if force: os.environ["CILQR_FORCE_G"] = str(force)
    else: os.environ.pop("CILQR_FORCE_G", None)
    p = cilqr_amd.default_params(N)
    sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x0, U, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
    pose, dim = (dv(sc["obs_pose"]), dv(sc["obs_dim"])) if M else (None, None)
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U0 = U.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        U.copy_(U0)
        torch.cuda.synchronize()
        e0.record()
        s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                             pose.data_ptr() if M else 0, dim.data_ptr() if M else 0, 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, float(it.float().mean())

shapes = [(30, 2), (50, 0), (50, 4), (50, 8), (50, 12), (80, 4), (80, 8), (80, 16), (100, 4), (120, 4), (120, 16), (160, 4)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in q.split("x")) for q in os.environ["SHAPES"].split(",")]
for N, M in shapes:
    for B in (2048, 4096, 8192, 16384):
        a, ia = run(N, M, B, 0)
        b, ib = run(N, M, B, 64)
        print("N=%3d M=%2d B=%5d  default %.3f ms | wavefront family %.3f ms  (mean iterations %.1f / %.1f)" % (N, M, B, a, b, ia, ib), flush=True)
