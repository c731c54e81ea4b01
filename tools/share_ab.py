#!/usr/bin/env python3
"""A/B of the shared-phase-L kernel (two wavefronts per solve working on phase L at the same time, cilqr_solve_share_kernel)
against the one-wavefront kernel on config-2 scenes: bit differences (there must be none) and kernel time by HIP events, over
a range of batch sizes.  Both handles solve the same inputs, taking turns in going first.

    [SHARE_W=2|3] python tools/share_ab.py [N] [M] [B ...]      default N = 50, M = 4, B = 1 64 256 512 768 1024 1536 2048
    (SHARE_W fixes the number of wavefronts per solve; default: the library's rule — three up to one solve per SIMD, else two)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BS = [int(v) for v in sys.argv[3:]] or [1, 64, 256, 512, 768, 1024, 1536, 2048]
REPS = 16
p = cilqr_amd.default_params(N)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
print("config-2 scenes, N = %d, M = %d static obstacles; kernel pair (fast + GENERAL) by HIP events, min / median of %d" % (N, M, REPS))
for B in BS:
    sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
    os.environ["CILQR_SHARE_MAX_B"] = str(1 << 30)  # the shared kernel at every batch size of this sweep
    if os.environ.get("SHARE_W"):
        os.environ["CILQR_SHARE_W"] = os.environ["SHARE_W"]
    two = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    del os.environ["CILQR_SHARE_MAX_B"]
    os.environ.pop("CILQR_SHARE_W", None)
    os.environ["CILQR_NO_SHARE_KERNEL"] = "1"
    one = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1))
    del os.environ["CILQR_NO_SHARE_KERNEL"]
    x0, U0, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
    pose, dim = (dv(sc["obs_pose"]), dv(sc["obs_dim"])) if M else (None, None)
    out = {}
    bufs = {}
    for name in ("two", "one"):
        bufs[name] = dict(X=torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"), J=torch.zeros(B, dtype=torch.float64, device="cuda"),
                          it=torch.zeros(B, dtype=torch.int32, device="cuda"), st=torch.zeros(B, dtype=torch.int32, device="cuda"), U=U0.clone(), ms=[])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(REPS + 2):
        for name in (("two", "one") if r % 2 == 0 else ("one", "two")):
            s, q = (two if name == "two" else one), bufs[name]
            q["U"].copy_(U0)
            torch.cuda.synchronize(); e0.record()
            s.solve_batch_device(torch.cuda.current_stream().cuda_stream, B, N, M, x0.data_ptr(), q["U"].data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                 pose.data_ptr() if M else 0, dim.data_ptr() if M else 0, 0, q["X"].data_ptr(), q["J"].data_ptr(), q["it"].data_ptr(), q["st"].data_ptr())
            e1.record(); torch.cuda.synchronize()
            if r >= 2:
                q["ms"].append(e0.elapsed_time(e1))
    a, b = bufs["two"], bufs["one"]
    same = all(torch.equal(a[k], b[k]) for k in ("U", "X", "J", "it", "st"))
    ta, tb = sorted(a["ms"]), sorted(b["ms"])
    print("B = %5d: %d wavefronts %.4f / %.4f ms | one %.4f / %.4f ms | ratio of medians %.3f | bit-identical: %s | families %s / %s"
          % (B, two.solve_wavefronts(B, N, M), ta[0], ta[len(ta) // 2], tb[0], tb[len(tb) // 2], ta[len(ta) // 2] / tb[len(tb) // 2], same,
             two.solve_family(B, N, M), one.solve_family(B, N, M)), flush=True)
    if not same:
        for k in ("U", "X", "J"):
            d = (a[k] - b[k]).abs()
            print("    %s: max |d| = %.3e; iters equal %s" % (k, float(d.max()), torch.equal(a["it"], b["it"])))
    two.close(); one.close()
