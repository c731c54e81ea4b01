#!/usr/bin/env python3
"""Does the order of the solves in a batch matter?  Static config-2 scenes and the config-3 scenes, timed as given and sorted by
the number of passes each solve runs (longest first / shortest first).  Diagnostic tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

def timed(fn, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

def static(B, N, M):
    p = cilqr_amd.default_params(N)
    sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    passes = torch.zeros(B, dtype=torch.int32, device="cuda")
    s.set_pass_count_buffer(passes.data_ptr())
    def run(perm):
        x0, U0, poly, xpl = dv(sc["x0"][perm]), dv(sc["U"][perm]), dv(sc["poly"][perm]), dv(sc["xplan_fl"][perm])
        pose, dim = dv(sc["obs_pose"][perm]), dv(sc["obs_dim"][perm])
        X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
        it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        U = U0.clone()
        def go():
            U.copy_(U0)
            s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                 pose.data_ptr(), dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        t = timed(go)
        return t, passes.cpu().numpy().copy()
    ident = np.arange(B)
    t0, ps = run(ident)
    t1, _ = run(np.argsort(-ps, kind="stable"))
    t2, _ = run(np.argsort(ps, kind="stable"))
    print("static N=%d M=%d B=%d: as given %.3f ms | longest first %.3f ms | shortest first %.3f ms  (passes mean %.1f max %d)" % (N, M, B, t0, t1, t2, ps.mean(), ps.max()), flush=True)

def c3(B):
    N = 50
    p = cilqr_amd.default_params(N)
    sc = scenes.make_c3(B, p)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=256)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    passes = torch.zeros(B, dtype=torch.int32, device="cuda")
    s.set_pass_count_buffer(passes.data_ptr())
    def run(perm):
        x0, U0, poly, xpl = dv(sc["x0"][perm]), dv(sc["U"][perm]), dv(sc["poly"][perm]), dv(sc["xplan_fl"][perm])
        pose, dim, off = dv(sc["nom_pose"][perm]), dv(sc["nom_dim"][perm]), dv(sc["offsets"][perm])
        X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
        it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        U = U0.clone()
        def go():
            U.copy_(U0)
            s.solve_batch_sampled_device(torch.cuda.current_stream().cuda_stream, B, N, 8, 32, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                         pose.data_ptr(), dim.data_ptr(), off.data_ptr(), 1.0 / 32, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        t = timed(go, 3)
        return t, passes.cpu().numpy().copy()
    ident = np.arange(B)
    t0, ps = run(ident)
    t1, _ = run(np.argsort(-ps, kind="stable"))
    t2, _ = run(np.argsort(ps, kind="stable"))
    print("config 3 B=%d: as given %.3f ms | longest first %.3f ms | shortest first %.3f ms  (passes mean %.1f max %d)" % (B, t0, t1, t2, ps.mean(), ps.max()), flush=True)

for B in (1024, 2048, 4096):
    static(B, 50, 4)
c3(4096)
