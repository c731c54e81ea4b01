#!/usr/bin/env python3
"""Grouped family with and without phase L's lane sharing (CILQR_NO_LANE_SHARING): kernel time by HIP events, best of 5.
    python tools/lane_sharing_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

def run(N, M, B, share, force=None):
    for k, v in (("CILQR_NO_LANE_SHARING", None if share else "1"), ("CILQR_FORCE_G", force)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    p = cilqr_amd.default_params(N)
    sc = run.cache.get((N, M, B)) or scenes.make_static(B, N, M, p, scenes.SEED0 + (5 if N == 80 else 2))
    run.cache = {(N, M, B): sc}
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x0, U0, poly, xpl, pose, dim = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"]), dv(sc["obs_pose"]), dv(sc["obs_dim"])
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U = U0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        U.copy_(U0); torch.cuda.synchronize(); e0.record()
        s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                             pose.data_ptr(), dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    g = s.solve_family(B, N, M)
    s.close()
    return best, g, U.cpu().numpy()
run.cache = {}

for N, M, B, force in ((80, 16, 8192, None), (80, 16, 4096, None), (80, 4, 16384, None), (50, 4, 16384, None), (50, 4, 32768, None), (50, 4, 65536, None), (50, 4, 8192, 8), (50, 4, 65536, 2), (50, 4, 65536, 4)):
    a, g, Ua = run(N, M, B, True, force)
    b, _, Ub = run(N, M, B, False, force)
    print("N=%d M=%d B=%d (G=%d): lane sharing %.3f ms (%.2f M solves/s) | without %.3f ms (%.2f M solves/s) | same bits: %s"
          % (N, M, B, g, a, B / a / 1e3, b, B / b / 1e3, np.array_equal(Ua, Ub)), flush=True)
