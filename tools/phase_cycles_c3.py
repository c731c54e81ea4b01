#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the wavefront family on config 3 in compact sampled form (diagnostic instantiation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
B, N = int(os.environ.get("B", 1024)), 50
p = cilqr_amd.default_params(N)
sc = scenes.make_c3(B, p)
s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=256)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
pose, dim, off = dv(sc["nom_pose"]), dv(sc["nom_dim"]), dv(sc["offsets"])
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
diag = torch.zeros(B, 16, dtype=torch.int64, device="cuda")
s.set_diag_buffer(diag.data_ptr())
s.solve_batch_sampled_device(0, B, N, 8, 32, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(),
                             dim.data_ptr(), off.data_ptr(), 1.0 / 32, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
torch.cuda.synchronize()
d = diag.cpu().numpy().astype(np.float64)
nL, nR = d[:, 5], d[:, 6]
print("config 3 sampled, solves", B)
print("per-solve totals (cycles): mean %.0f  max %.0f" % (d[:, 7].mean(), d[:, 7].max()))
print("L per call: %.0f  (calls mean %.1f max %d)" % ((d[:, 1] / nL).mean(), nL.mean(), nL.max()))
m = nR > 0
print("R per call: %.0f | F per call: %.0f" % ((d[m, 2] / nR[m]).mean(), (d[m, 3] / nR[m]).mean()))
