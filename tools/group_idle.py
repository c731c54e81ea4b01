#!/usr/bin/env python3
"""How much of a grouped-family wavefront idles: executed passes per solve (cilqr_set_pass_count_buffer) against the maximum
over the S = 64/G solves that share a wavefront and stay in its wave-uniform loop until the last one has finished.
Diagnostic tool.   python tools/group_idle.py [c5|c2] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

kind = sys.argv[1] if len(sys.argv) > 1 else "c5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N, M = (80, 16) if kind == "c5" else (50, 4)
p = cilqr_amd.default_params(N)
sc = scenes.make_c5(B, p) if kind == "c5" else scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U, poly, xpl, pose, dim = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"]), dv(sc["obs_pose"]), dv(sc["obs_dim"])
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
ps = torch.zeros(B, dtype=torch.int32, device="cuda")
s.set_pass_count_buffer(ps.data_ptr())
s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                     pose.data_ptr(), dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
torch.cuda.synchronize()
passes, status = ps.cpu().numpy(), st.cpu().numpy()
trips = passes + (status != 0)  # loop trips a solve needs: every pass, plus the linearisation + backward pass of the rejected iteration
G = s.solve_family(B, N, M)
print("%s B=%d N=%d M=%d: family G=%d; passes mean %.2f max %d; loop trips per solve mean %.2f" % (kind, B, N, M, G, passes.mean(), passes.max(), trips.mean()))
print("histogram of trips:", np.bincount(trips, minlength=22).tolist())
for g in (32, 16, 8, 4, 2, 1):
    S = 64 // g
    n = B // S * S
    w = trips[:n].reshape(-1, S).max(axis=1)
    print("  S = %2d solves per wavefront (G = %2d): trips of a wavefront mean %.2f (max %d) -> lane-groups busy %.0f %% of its loop"
          % (S, g, w.mean(), w.max(), 100 * trips[:n].mean() / w.mean()))
