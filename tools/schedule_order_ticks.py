#!/usr/bin/env python3
"""The schedule hint on batches that CHANGE from call to call.

The one-wavefront-per-solve family dispatches a batch of more solves than SIMDs longest-first by the pass counts of the PREVIOUS
call.  bench.py used to repeat one batch — the hint's best case.  Here the batch is a closed-loop sequence of
planner ticks (cilqr_amd.scenes.TickSequence: ego moved one step along the accepted plan, warm-started un-shifted controls,
re-fitted local plan, obstacles moved on by one timestep, fresh pose noise of the launch-file sigmas), and every tick is solved
twice on the same inputs: on a handle with the hint and on one created with CILQR_NO_SCHEDULE_HINT.  Kernel time by HIP events
around the launch; the results are bit-identical in any dispatch order (checked here on every tick).

    python tools/schedule_order_ticks.py [ticks]     # config 3 (B = 4096) and config-2 scenes at B = 4096 / 2048
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cilqr_amd  # noqa: E402
from cilqr_amd import scenes  # noqa: E402


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sequence(kind, B, ticks, N=50, M=4):
    p = cilqr_amd.default_params(N)
    ts = scenes.TickSequence(kind, B, p, N=N, M=M)
    Mmax = 256 if kind == "c3" else M
    hinted = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=Mmax)
    os.environ["CILQR_NO_SCHEDULE_HINT"] = "1"
    plain = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=Mmax)
    del os.environ["CILQR_NO_SCHEDULE_HINT"]
    passes = torch.zeros(B, dtype=torch.int32, device="cuda")
    hinted.set_pass_count_buffer(passes.data_ptr())
    X = [torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda") for _ in range(2)]
    J = [torch.zeros(B, dtype=torch.float64, device="cuda") for _ in range(2)]
    it = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    st = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    prev_passes = None
    for t in range(ticks):
        i = ts.inputs()
        x0, U0, poly, xpl = dv(i["x0"]), dv(i["U"]), dv(i["poly"]), dv(i["xplan_fl"])
        if kind == "c3":
            pose, dim, off = dv(i["nom_pose"]), dv(i["nom_dim"]), dv(i["offsets"])
        else:
            pose, dim = dv(i["obs_pose"]), dv(i["obs_dim"])
        ms, Us = [None, None], [None, None]
        # the two handles take turns in going first (whoever runs second finds the tick's inputs in the caches)
        for k in ((0, 1) if t % 2 == 0 else (1, 0)):
            slv = (hinted, plain)[k]
            U = U0.clone()
            torch.cuda.synchronize()
            e0.record()
            if kind == "c3":
                slv.solve_batch_sampled_device(stream, B, N, ts.n_obs, i["offsets"].shape[2], x0.data_ptr(), U.data_ptr(), poly.data_ptr(),
                                               xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), off.data_ptr(), i["sample_weight"],
                                               X[k].data_ptr(), J[k].data_ptr(), it[k].data_ptr(), st[k].data_ptr())
            else:
                slv.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(),
                                       dim.data_ptr(), 0, X[k].data_ptr(), J[k].data_ptr(), it[k].data_ptr(), st[k].data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms[k] = e0.elapsed_time(e1)
            Us[k] = U
        assert torch.equal(Us[0], Us[1]) and torch.equal(X[0], X[1]) and torch.equal(it[0], it[1]), "dispatch order changed a result"
        ps = passes.cpu().numpy().copy()
        # how well the previous tick's pass counts predict this tick's (what the hint relies on)
        corr = float(np.corrcoef(prev_passes, ps)[0, 1]) if prev_passes is not None and ps.std() > 0 and prev_passes.std() > 0 else float("nan")
        prev_passes = ps
        rows.append((t, ms[0], ms[1], float(ps.mean()), int(ps.max()), float(it[0].float().mean().item()), corr))
        print("  tick %2d: with hint %.3f ms | without %.3f ms | passes mean %.2f max %d | reference iterations mean %.2f | "
              "pass-count correlation with the previous tick %.2f" % rows[-1], flush=True)
        ts.advance(X[0].cpu().numpy(), Us[0].cpu().numpy())
    hinted.close()
    plain.close()
    a = np.array([(r[1], r[2]) for r in rows[1:]])  # tick 0 has no hint yet
    first = np.array([(r[1], r[2]) for r in rows[1:] if r[0] % 2 == 0]), np.array([(r[1], r[2]) for r in rows[1:] if r[0] % 2 == 1])
    print("  (hinted handle first: with %.3f / without %.3f ms; plain handle first: with %.3f / without %.3f ms)"
          % (first[0][:, 0].mean(), first[0][:, 1].mean(), first[1][:, 0].mean(), first[1][:, 1].mean()))
    print("%s B=%d, ticks 1..%d: with hint %.3f ms mean (%.2f M solves/s) | without %.3f ms mean (%.2f M solves/s) | first tick %.3f ms"
          % (kind, B, ticks - 1, a[:, 0].mean(), B / a[:, 0].mean() / 1e3, a[:, 1].mean(), B / a[:, 1].mean() / 1e3, rows[0][1]), flush=True)


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    print("config 3: B = 4096, N = 50, 8 moving obstacles x 32 samples, compact form")
    sequence("c3", 4096, T)
    for B in (4096, 2048):
        print("config-2 scenes: B = %d, N = 50, M = 4 static obstacles" % B)
        sequence("static", B, T)
