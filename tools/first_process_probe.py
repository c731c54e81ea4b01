#!/usr/bin/env python3
"""Is this process in the alternating state (launch sequence)?  K back-to-back launches of config 2, HIP events around each;
prints mean and the even / odd launch means per block of 200.  Seen so far: the first process on one fresh box alternated for its whole
life (3000 launches: 0.413 / 0.386 ms) and the next one did not; bench.py runs right behind a rocprofv3 --pmc session alternated (2.41 M
solves/s) and the identical run behind them did not (2.48 M), four times out of four; kernel arguments forced into host memory
(HIP_FORCE_DEV_KERNARG=0) cost 4.5 µs per launch and no alternation.   python tools/first_process_probe.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
B, N, M = 1024, 50, 4
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
p = cilqr_amd.default_params(N)
sc = scenes.make_c2(B, p)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U0, poly, xpl, pose, dim = (dv(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"))
s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
U = U0.clone()
stream = torch.cuda.current_stream().cuda_stream
e0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
e1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
for k in range(K):
    U.copy_(U0)
    e0[k].record()
    s.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                         X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
    e1[k].record()
torch.cuda.synchronize()
d = np.array([a.elapsed_time(b) for a, b in zip(e0, e1)])
tag = os.environ.get("PROBE_TAG", "")
for c in range(0, K, 200):
    w = d[c:c + 200]
    print("%s launches %4d-%4d: mean %.4f ms  even / odd %.4f / %.4f  min %.4f" % (tag, c, c + len(w) - 1, w.mean(), w[0::2].mean(), w[1::2].mean(), w.min()))
s.close()
