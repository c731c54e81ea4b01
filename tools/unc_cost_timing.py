#!/usr/bin/env python3
"""What the costmap-lookup uncertainty term costs a solve: kernel time (HIP events) of config-2 scenes with and without a map set
(3 × 3 footprint probes, shared map), at several batch sizes.   python tools/unc_cost_timing.py [B ...]
With SAMPLED=1 in the environment: config-3 scenes (8 moving obstacles × 32 pose samples, compact form) instead, on the split kernel and,
with CILQR_NO_SPLIT_KERNEL, on one wavefront per solve."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes
from oracle import oracle as O

BS = [int(v) for v in sys.argv[1:]] or [1, 256, 1024]
N, M = 50, 4
p = cilqr_amd.default_params(N)
p.safe_length, p.safe_width = 1.1, 0.9
geom = (30.0, 20.0, 0.2, 15.0, 0.0)
og = O.map_geom(*geom)
occ = scenes.make_occupancy(og.rows, og.cols, 91)
layer, _, _ = O.blur(np.nan_to_num(occ, nan=0.0), og, np.sin(0.1), np.cos(0.1), 0.16, 0.16, 0.017, threads=8)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
SAMPLED = bool(os.environ.get("SAMPLED"))
if SAMPLED:
    for B in BS:
        sc = scenes.make_c3(B, p)
        for split in (True, False):
            if not split:
                os.environ["CILQR_NO_SPLIT_KERNEL"] = "1"
            s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=256)
            os.environ.pop("CILQR_NO_SPLIT_KERNEL", None)
            x0, U0, poly, xpl, pose, dim, off = (dv(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "nom_pose", "nom_dim", "offsets"))
            X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
            it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
            U = U0.clone()
            stream = torch.cuda.current_stream().cuda_stream
            res = {}
            for name in ("no map", "map set"):
                if name == "map set":
                    s.set_uncertainty_map(layer, cilqr_amd.map_geom(*geom), (-1.0, 0.4, 0.05), (3, 3))
                ts = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for r in range(6):
                    U.copy_(U0); torch.cuda.synchronize(); e0.record()
                    s.solve_batch_sampled_device(stream, B, N, 8, 32, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(),
                                                 off.data_ptr(), sc["sample_weight"], X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
                    e1.record(); torch.cuda.synchronize()
                    if r >= 1: ts.append(e0.elapsed_time(e1))
                res[name] = (min(ts), s.solve_sampled_wavefronts(B, N, 8))
            print("config-3 scenes, B = %5d, %s: no map %.3f ms (%d wavefront(s) per solve) | map set %.3f ms (%d wavefront(s))"
                  % (B, "split kernel" if split else "one wavefront per solve", res["no map"][0], res["no map"][1], res["map set"][0], res["map set"][1]), flush=True)
            s.close()
    sys.exit(0)
for B in BS:
    sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
    s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
    x0, U0, poly, xpl, pose, dim = (dv(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"))
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    U = U0.clone()
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for name in ("no map", "map set"):
        if name == "map set":
            s.set_uncertainty_map(layer, cilqr_amd.map_geom(*geom), (-1.0, 0.4, 0.05), (3, 3))
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(12):
            U.copy_(U0); torch.cuda.synchronize(); e0.record()
            s.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                                 X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
            e1.record(); torch.cuda.synchronize()
            if r >= 2: ts.append(e0.elapsed_time(e1))
        out[name] = (min(ts), float(it.float().mean()), s.solve_wavefronts(B, N, M))
    print("B = %5d: no map %.4f ms (mean iterations %.1f, %d wavefront(s) per solve) | map set %.4f ms (mean iterations %.1f, %d wavefront(s))"
          % (B, out["no map"][0], out["no map"][1], out["no map"][2], out["map set"][0], out["map set"][1], out["map set"][2]), flush=True)
    s.close()
