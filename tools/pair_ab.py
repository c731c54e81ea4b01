#!/usr/bin/env python3
"""A/B of the two-wavefront kernel against the one-wavefront kernel on one batch: results (bit differences and their size) and
kernel time by HIP events.  Diagnostic tool.   python tools/pair_ab.py [B] [N] [M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p = cilqr_amd.default_params(N)
sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
os.environ["CILQR_PAIR_KERNEL"] = "1"
two = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
del os.environ["CILQR_PAIR_KERNEL"]
one = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U0, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
pose, dim = (dv(sc["obs_pose"]), dv(sc["obs_dim"])) if M else (None, None)
out = {}
for name, s in (("two", two), ("one", one)):
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U = U0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(12):
        U.copy_(U0)
        torch.cuda.synchronize(); e0.record()
        s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                             pose.data_ptr() if M else 0, dim.data_ptr() if M else 0, 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out[name] = dict(U=U.cpu().numpy(), X=X.cpu().numpy(), J=J.cpu().numpy(), it=it.cpu().numpy(), st=st.cpu().numpy(), ms=sorted(ts))
    print("%s wavefront(s) per solve: kernel pair %.4f ms min, %.4f median" % (name, ts and min(ts), sorted(ts)[len(ts) // 2]))
a, b = out["two"], out["one"]
print("iters equal:", np.array_equal(a["it"], b["it"]), " status equal:", np.array_equal(a["st"], b["st"]))
for k in ("U", "X", "J"):
    d = np.abs(a[k] - b[k])
    nd = int(np.sum(np.any(np.atleast_2d(d.reshape(B, -1) != 0), axis=1)))
    print("%s: %d of %d solves differ, max |d| = %.3e, max rel = %.3e" % (k, nd, B, np.nanmax(d), np.nanmax(d / (1e-300 + np.abs(b[k])))))
