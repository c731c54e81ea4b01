#!/usr/bin/env python3
"""The durations of K launches of one config-2 batch, HIP events around every launch: enqueued back to back (no host synchronisation
in between — as bench.py enqueues its steps) and with a synchronisation after every launch.  Back to back, every other launch of a
batch that fills the chip takes ≈ 25 µs longer (MI355X, round 3; the placement of the wavefronts is the same in both:
tools/wave_placement.py); synchronised, all take the shorter time.

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
solver = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
# mode: (synchronise after every launch, extra trivial dispatches enqueued before every launch)
handles = {"back to back": (False, 0), "synchronised": (True, 0), "one extra dispatch": (False, 1), "two extra dispatches": (False, 2)}
dummy = torch.zeros(64, device="cuda")
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
U = U0.clone()
stream = torch.cuda.current_stream().cuda_stream
print("config-2 scenes, B = %d, %d wavefronts per solve; %d launches per run, three runs per mode, in turns" % (B, solver.solve_wavefronts(B, N, M), K))
for rep in range(3):
    for name, (sync, extra) in handles.items():
        s = solver
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        torch.cuda.synchronize()
        for k in range(K):
            U.copy_(U0)
            for _ in range(extra):
                dummy.add_(1.0)
            e0[k].record()
            s.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                                 X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
            e1[k].record()
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        d = np.array([a.elapsed_time(b) for a, b in zip(e0, e1)])[4:]
        print("%-20s run %d: min %.4f  median %.4f  mean %.4f  max %.4f ms | launches above 1.05 x min: %d of %d | first 16: %s"
              % (name, rep, d.min(), np.median(d), d.mean(), d.max(), int((d > 1.05 * d.min()).sum()), len(d), " ".join("%.0f" % (1e3 * v) for v in d[:16])), flush=True)
# The stamped instantiation in the same process: duration by events and the longest solve's length in shader clocks, launch by launch —
# equal clocks with alternating durations mean the CLOCK alternates, not the work.
diag = [torch.zeros(B, 16, dtype=torch.int64, device="cuda") for _ in range(16)]
e0 = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
e1 = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
torch.cuda.synchronize()
for k in range(16):
    solver.set_diag_buffer(diag[k].data_ptr())
    U.copy_(U0)
    e0[k].record()
    solver.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                              X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
    e1[k].record()
torch.cuda.synchronize()
print("stamped instantiation, 16 launches: ms | longest solve in shader ticks | ticks per ns")
for k in range(16):
    ms, ticks = e0[k].elapsed_time(e1[k]), int(diag[k][:, 7].max().item())
    print("   launch %2d: %.4f ms | %d | %.3f" % (k, ms, ticks, ticks / (ms * 1e6)))
solver.close()
