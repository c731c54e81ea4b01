#!/usr/bin/env python3
"""Config-3 scenes: two wavefronts per solve sharing phase L (default) against one (CILQR_NO_SPLIT_KERNEL), first calls without
the schedule hint.  python tools/split_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
os.environ["CILQR_NO_SCHEDULE_HINT"] = "1"
N = 50
p = cilqr_amd.default_params(N)
for B in (256, 1024, 2048, 4096, 8192):
    sc = scenes.make_c3(B, p)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x0, U0, poly, xpl, pose, dim, off = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"]), dv(sc["nom_pose"]), dv(sc["nom_dim"]), dv(sc["offsets"])
    res = {}
    for name, env, w in (("split", None, "2"), ("split4", None, "4"), ("one", "1", "2")):
        os.environ["CILQR_SPLIT_W"] = w
        if env: os.environ["CILQR_NO_SPLIT_KERNEL"] = env
        else: os.environ.pop("CILQR_NO_SPLIT_KERNEL", None)
        s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=256)
        X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
        it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        U = U0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(4):
            U.copy_(U0); torch.cuda.synchronize(); e0.record()
            s.solve_batch_sampled_device(torch.cuda.current_stream().cuda_stream, B, N, 8, 32, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                         pose.data_ptr(), dim.data_ptr(), off.data_ptr(), 1.0 / 32, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = (min(ts), U.cpu().numpy(), it.cpu().numpy())
        s.close()
    a, a4, b = res["split"], res["split4"], res["one"]
    print("config-3 scenes B=%d: two wavefronts per solve %.3f ms (%.2f M solves/s) | four %.3f ms (%.2f M solves/s) | one %.3f ms (%.2f M solves/s) | iterations equal %s, max|dU| %.2e / %.2e"
          % (B, a[0], B / a[0] / 1e3, a4[0], B / a4[0] / 1e3, b[0], B / b[0] / 1e3, np.array_equal(a[2], b[2]) and np.array_equal(a4[2], b[2]),
             np.abs(a[1] - b[1]).max(), np.abs(a4[1] - b[1]).max()), flush=True)
