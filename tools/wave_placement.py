#!/usr/bin/env python3
"""Where the wavefronts of the shared-phase-L kernel run: HW_ID / XCC_ID of every wavefront of every solve (stamped instantiation),
over K back-to-back launches of one config-2 batch.  Per launch: its duration (HIP events), how many SIMDs hold two MAIN wavefronts,
how many workgroups have both their wavefronts on one SIMD, workgroups per CU, and who shares a SIMD with the slowest solves.

    python tools/wave_placement.py [B] [K]
"""
import os, sys
from collections import Counter, defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")]
import numpy as np, torch
import cilqr_amd
from cilqr_amd import scenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, M = 50, 4
p = cilqr_amd.default_params(N)
sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, U0, poly, xpl, pose, dim = (dv(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"))
s = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
W = s.solve_wavefronts(B, N, M)
X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device="cuda"); J = torch.zeros(B, dtype=torch.float64, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
diags = [torch.zeros(B, 16, dtype=torch.int64, device="cuda") for _ in range(K)]
U = U0.clone()
stream = torch.cuda.current_stream().cuda_stream
e0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
e1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
for k in range(K):
    s.set_diag_buffer(diags[k].data_ptr())
    U.copy_(U0)
    e0[k].record()
    s.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(), dim.data_ptr(), 0,
                         X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
    e1[k].record()
torch.cuda.synchronize()


def where(v):  # (xcc, se, sh, cu, simd), wave slot
    v = int(v)
    hw, xcc = v & 0xFFFFFFFF, (v >> 32) & 0xF
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3), hw & 15


print("config-2 scenes, B = %d, %d wavefronts per solve (stamped instantiation), %d back-to-back launches" % (B, W, K))
for k in range(K):
    d = diags[k].cpu().numpy()
    total = d[:, 7]
    simd_main, simd_all, cu_wg = defaultdict(list), defaultdict(list), Counter()
    same = 0
    for b in range(B):
        m, _ = where(d[b, 12])
        simd_main[m].append(b)
        simd_all[m].append((b, "main"))
        cu_wg[m[:4]] += 1
        for w in range(1, W):
            a, _ = where(d[b, 12 + w])
            simd_all[a].append((b, "aux%d" % w))
            same += a == m
    two_mains = sum(1 for v in simd_main.values() if len(v) >= 2)
    waves_per_simd = Counter(len(v) for v in simd_all.values())
    slow = np.argsort(-total)[:3]
    desc = []
    for b in slow:
        m, _ = where(d[b, 12])
        others = [("%s of %d (%d passes)" % (r, o, d[o, 6])) for o, r in simd_all[m] if o != b]
        desc.append("solve %d (%d passes, %d ticks) shares its SIMD with: %s" % (b, d[b, 6], total[b], ", ".join(others) or "nobody"))
    print("launch %d: %.4f ms | CUs used %d, workgroups per CU %s | SIMDs used %d, wavefronts per SIMD %s | SIMDs with two or more main wavefronts: %d | "
          "workgroups with a second wavefront on main's SIMD: %d" % (k, e0[k].elapsed_time(e1[k]), len(cu_wg), dict(sorted(Counter(cu_wg.values()).items())),
                                                                   len(simd_all), dict(sorted(waves_per_simd.items())), two_mains, same))
    for t in desc:
        print("      " + t)
s.close()
