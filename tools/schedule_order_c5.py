#!/usr/bin/env python3
"""Grouped family and the order of the solves: config-5 scenes (B = 8192, N = 80, M = 16) as given and sorted by pass count, at the
automatic G = 8 (one wavefront per SIMD) and at G = 16 and 32 forced (two and four wavefronts per SIMD).  Diagnostic tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
B, N, M = 8192, 80, 16
p = cilqr_amd.default_params(N)
sc = scenes.make_c5(B, p, shard=0)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def run(G, perm):
    if G: os.environ["CILQR_FORCE_G"] = str(G)
    else: os.environ.pop("CILQR_FORCE_G", None)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
    passes = torch.zeros(B, dtype=torch.int32, device="cuda")
    s.set_pass_count_buffer(passes.data_ptr())
    x0, U0, poly, xpl = dv(sc["x0"][perm]), dv(sc["U"][perm]), dv(sc["poly"][perm]), dv(sc["xplan_fl"][perm])
    pose, dim = dv(sc["obs_pose"][perm]), dv(sc["obs_dim"][perm])
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U = U0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        U.copy_(U0); torch.cuda.synchronize(); e0.record()
        s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                             pose.data_ptr(), dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    ps = passes.cpu().numpy().copy()
    s.close()
    return best, ps
ident = np.arange(B)
t8, ps = run(0, ident)
srt = np.argsort(-ps, kind="stable")
print("passes mean %.1f max %d" % (ps.mean(), ps.max()))
for G in (0, 16, 32):
    a, _ = run(G, ident)
    b, _ = run(G, srt)
    print("G=%s: as given %.3f ms | longest first %.3f ms" % (G or "auto(8)", a, b), flush=True)
