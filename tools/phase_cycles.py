#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the solve kernel on config 2 (diagnostic instantiation; not a timing run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
FLAGS = int(os.environ.get("FLAGS", 0))
B, N, M = int(os.environ.get("B", 1024)), int(os.environ.get("N", 50)), int(os.environ.get("M", 4))
p = cilqr_amd.default_params(N)
sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
pose, dim = (dv(sc["obs_pose"]), dv(sc["obs_dim"])) if M else (None, None)
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
diag = torch.zeros(B, 16, dtype=torch.int64, device="cuda")
s.set_diag_buffer(diag.data_ptr())
U0 = U.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):  # second run timed (same stamped kernel)
    U.copy_(U0)
    torch.cuda.synchronize()
    e0.record()
    s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr() if M else 0,
                         dim.data_ptr() if M else 0, 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr(), FLAGS)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
U.copy_(U0)
s.solve_batch_device(0, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr() if M else 0,
                     dim.data_ptr() if M else 0, 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr(), FLAGS)
torch.cuda.synchronize()
d = diag.cpu().numpy().astype(np.float64)
nL, nR = d[:, 5], d[:, 6]
print("solves", B, "N", N, "M", M)
print("stamped launch: %.3f ms; longest solve %.0f ticks -> %.2f ticks/ns (shader clock if the counter is the shader clock)" % (ms, d[:, 7].max(), d[:, 7].max() / (ms * 1e6)))
print("per-solve totals (cycles): mean %.0f  max %.0f" % (d[:, 7].mean(), d[:, 7].max()))
print("prologue mean %.0f | epilogue mean %.0f" % (d[:, 0].mean(), d[:, 4].mean()))
print("L per call: %.0f  (calls mean %.1f max %d)" % ((d[:, 1] / nL).mean(), nL.mean(), nL.max()))
m = nR > 0
print("R per call: %.0f -> %.0f per step" % ((d[m, 2] / nR[m]).mean(), (d[m, 2] / nR[m]).mean() / N))
print("F per call: %.0f -> %.0f per step" % ((d[m, 3] / nR[m]).mean(), (d[m, 3] / nR[m]).mean() / N))
w = int(np.argmax(d[:, 7]))
print("slowest solve %d: pro %d L %d R %d F %d epi %d nL %d nR %d total %d" % ((w,) + tuple(int(v) for v in d[w][:8])))
if os.environ.get("SHARE"):  # shared-phase-L kernel: slots 8..10 = aux busy ticks, aux calls, main's wait at barrier A
    k = np.maximum(d[:, 9], 1)
    print("aux wavefront: %.0f ticks per linearisation (calls mean %.1f) | second aux wavefront %.0f | main waits %.0f ticks per linearisation at barrier A"
          % ((d[:, 8] / k).mean(), d[:, 9].mean(), (d[:, 11] / k).mean(), (d[:, 10] / nL).mean()))
    w = int(np.argmax(d[:, 7]))
    print("slowest solve %d: aux %.0f / %.0f ticks per linearisation, main waits %.0f" % (w, d[w, 8] / k[w], d[w, 11] / k[w], d[w, 10] / nL[w]))
elif os.environ.get("PAIR"):  # two-wavefront kernel: slots 8..11 are the aux wavefront's account
    k = d[:, 11]
    print("aux wavefront, ticks per linearisation: waiting for states %.0f | chunk work %.0f | from the last state's arrival to its barrier %.0f  (calls mean %.1f)"
          % ((d[:, 8] / k).mean(), (d[:, 9] / k).mean(), (d[:, 10] / k).mean(), k.mean()))
elif d[:, 8:].sum() > 0:  # sub-phase stamps (one-wavefront-per-solve family)
    names = ["cos/sin columns", "closest sample (+ forward-record stores)", "cost derivatives (lin_step)", "record stores (+ map term)", "cost reduction"]
    print("inside L, ticks per call:  " + "  ".join("%s %.0f" % (n, (d[:, 8 + i] / nL).mean()) for i, n in enumerate(names)))
    names = ["P, Db, Da + readlanes", "determinant + reciprocal", "Dk, H, Dv, copies, stores"]
    print("inside R, ticks per step:  " + "  ".join("%s %.0f" % (n, (d[m, 13 + i] / nR[m]).mean() / N) for i, n in enumerate(names))
          + "   (four s_memtime and one s_waitcnt lgkmcnt(0) per step: compare the sum with the unstamped R per step of a DIAG-free run)")
