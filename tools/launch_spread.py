#!/usr/bin/env python3
"""How evenly a batch that fits the chip at once lies on it: the durations of K back-to-back launches of one config-2 batch (HIP
events around every launch, no host synchronisation in between — as bench.py enqueues them), on a handle with the balanced LDS
request (cilqr_api.cpp, balanced_lds_bytes) and on one created with CILQR_NO_LDS_BALANCE.

    python tools/launch_spread.py [B] [K]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, M = 50, 4
p = cilqr_amd.default_params(N)
sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U0, poly, xpl, pose, dim = (dv(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"))
handles = {}
handles["balanced"] = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
os.environ["CILQR_NO_LDS_BALANCE"] = "1"
handles["as placed"] = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
del os.environ["CILQR_NO_LDS_BALANCE"]
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
U = U0.clone()
stream = torch.cuda.current_stream().cuda_stream
print("config-2 scenes, B = %d, %d wavefronts per solve; %d back-to-back launches per run, three runs per handle, in turns" % (B, handles["balanced"].solve_wavefronts(B, N, M), K))
for rep in range(3):
    for name, s in handles.items():
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        torch.cuda.synchronize()
        for k in range(K):
            U.copy_(U0)
            e0[k].record()
            s.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                                 X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
            e1[k].record()
        torch.cuda.synchronize()
        d = np.array([a.elapsed_time(b) for a, b in zip(e0, e1)])[4:]
        print("%-10s run %d: min %.4f  median %.4f  mean %.4f  max %.4f ms | launches above 1.05 x min: %d of %d | first 16: %s"
              % (name, rep, d.min(), np.median(d), d.mean(), d.max(), int((d > 1.05 * d.min()).sum()), len(d), " ".join("%.0f" % (1e3 * v) for v in d[:16])), flush=True)
for s in handles.values():
    s.close()
