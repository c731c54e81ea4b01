import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/uncertainty-aware-cilqr-for-trajectory-optimization_amd")
from oracle import oracle as O
import cilqr_amd
from cilqr_amd import scenes
for name, kw in (("w_acc=-0.5", dict(w_acc=-0.5)), ("w_yaw=-1", dict(w_yawrate=-1.0)), ("w_acc=-2", dict(w_acc=-2.0)), ("both", dict(w_acc=-0.3, w_yawrate=-0.5))):
    p = cilqr_amd.default_params(50); po = O.default_params(50)
    for k, v in kw.items():
        setattr(p, k, v); setattr(po, k, v)
    sc = scenes.make_c2(16, p)
    s = cilqr_amd.Solver(p, max_batch=16, max_horizon=50, max_obstacles=4)
    g = s.solve_batch(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"])
    o = O.solve_batch(po, 50, 4, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, threads=8)
    du = np.max(np.abs(g["U"]-o["U"]),axis=1)
    print(name); print(" gpu it", g["iters"], "st", g["status"]); print(" ora it", o["iters"], "st", o["status"]); print(" du", np.array2string(du, precision=1))
    s.close()
