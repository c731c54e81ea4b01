"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on identical seeded inputs.

Tolerances (stated here, used below):
  * solver (fp64):  max|ΔU| ≤ 1e-6 is the BASELINE.json bar; TIGHT = 1e-9 is what this suite actually enforces on
    scenes that are not decision knife-edges (observed ≈1e-12).  Iteration counts and exit reasons must be EQUAL.
  * costmap warp (integer index math, float32 payload): bit-exact.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

BAR = 1e-6
TIGHT = 1e-9


@pytest.fixture(scope="module")
def solver(cilqr):
    p = cilqr.default_params()
    s = cilqr.Solver(p, max_batch=2048, max_horizon=80, max_obstacles=64, device=0)
    yield s
    s.close()


def _oracle_batch(O, N, sc, threads=16):
    p = O.default_params(N)
    return O.solve_batch(p, N, sc["M"], sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"],
                         sc["obs_weight"], threads=min(threads, O.max_threads()))


def _gpu_batch(solver, sc, flags=0):
    return solver.solve_batch(sc["N"], sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"],
                              sc["obs_weight"], flags=flags)


def _compare(got, want, tol, what=""):
    same_path = (got["iters"] == want["iters"]) & (got["status"] == want["status"])
    du = np.max(np.abs(got["U"] - want["U"]), axis=1)
    dx = np.max(np.abs(got["X"] - want["X"]), axis=1)
    assert same_path.all(), "%s: %d solves took a different accept/reject path" % (what, int((~same_path).sum()))
    assert du.max() <= tol, "%s: max|dU| = %g" % (what, du.max())
    assert dx.max() <= 100 * tol, "%s: max|dX| = %g" % (what, dx.max())
    assert np.allclose(got["J"], want["J"], rtol=1e-9, atol=1e-9), what
    return du.max()


@pytest.mark.parametrize("case", [c for c in load_golden("survey_known_answers.json")["cases"] if c["U0"] is not None],
                         ids=lambda c: "N%d_M%d" % (c["N"], c["M"]))
def test_known_answers_through_c_abi(cilqr, solver, case):
    """The SURVEY §8(c) reference outputs, reproduced by the HIP path itself."""
    from cilqr_amd import scenes
    N, M = case["N"], case["M"]
    sc = scenes.known_answer_scene(N, M, cilqr.default_params(N))
    r = _gpu_batch(solver, sc)
    assert r["iters"][0] == case["iterations"]
    assert r["status"][0] == {"lambda_max": 1, "max_iter": 2}[case["exit"]]
    assert np.max(np.abs(r["U"][0, :2] - np.array(case["U0"]))) < TIGHT
    assert np.max(np.abs(r["X"][0, -4:] - np.array(case["XN"]))) < 1e-8


def test_config2_batch_1024(cilqr, oracle, solver):
    """BASELINE config 2 at full size: B=1024, N=50, M=4."""
    from cilqr_amd import scenes
    sc = scenes.make_c2(1024, cilqr.default_params(50))
    got, want = _gpu_batch(solver, sc), _oracle_batch(oracle, 50, sc)
    worst = _compare(got, want, TIGHT, "C2")
    print("C2 B=1024 max|dU| = %.3e" % worst)
    assert worst <= BAR


def test_early_exit_equals_reference_loop(cilqr, solver):
    """CILQR_FLAG_FAITHFUL_ITERS runs the rejected iterations' passes as the reference loop does: bit-identical results."""
    from cilqr_amd import scenes
    sc = scenes.make_c2(256, cilqr.default_params(50))
    a, b = _gpu_batch(solver, sc), _gpu_batch(solver, sc, flags=cilqr.FLAG_FAITHFUL_ITERS)
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("G", [0, 8])
def test_general_only_flag(cilqr, oracle, monkeypatch, G):
    """CILQR_FLAG_GENERAL_ONLY hands every solve to the GENERAL kernels (library-range sincos, eigenvalue-clamping inverse, the value
    update as the reference's direct product): same accept / reject paths and results to rounding as the production kernels and the
    oracle, both families — the path that otherwise only runs for the rare solve the production kernels do not cover."""
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 50, 4, 192
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 8123)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        prod = _gpu_batch(s, sc)
        gen = _gpu_batch(s, sc, flags=cilqr.FLAG_GENERAL_ONLY)
    finally:
        s.close()
    _compare(gen, prod, 1e-10, "GENERAL kernels only against the production kernels G=%d" % G)
    _compare(gen, _oracle_batch(oracle, N, sc), TIGHT, "GENERAL kernels only G=%d" % G)


@pytest.mark.parametrize("N,M,B,seed", [(30, 2, 128, 101), (80, 16, 128, 102), (50, 0, 64, 103), (1, 1, 8, 104), (64, 3, 32, 105),
                                        (65, 3, 32, 106), (7, 5, 16, 107), (2, 1, 8, 108), (3, 2, 8, 109)])
def test_other_shapes(cilqr, oracle, solver, N, M, B, seed):
    """Config 1 / config 5 shapes, no obstacles, horizons around the wavefront width, tiny horizons."""
    from cilqr_amd import scenes
    sc = scenes.make_static(B, N, M, cilqr.default_params(N), seed)
    _compare(_gpu_batch(solver, sc), _oracle_batch(oracle, N, sc), TIGHT, "N%d M%d" % (N, M))


def test_config3_sampled_obstacles(cilqr, oracle, solver):
    """Config 3 shape at reduced batch: 8 moving obstacles × 8 Gaussian samples, weight 1/8 (per-obstacle weights)."""
    from cilqr_amd import scenes
    sc = scenes.make_c3(48, cilqr.default_params(50), n_dyn=8, n_samples=8)
    _compare(_gpu_batch(solver, sc), _oracle_batch(oracle, 50, sc), TIGHT, "C3")


def test_warm_start_second_tick(cilqr, oracle, solver):
    """control_seq persists un-shifted across run_step calls (I/iLQR.cpp:253): feed U_result back in."""
    from cilqr_amd import scenes
    sc = scenes.make_c2(64, cilqr.default_params(50))
    g1, o1 = _gpu_batch(solver, sc), _oracle_batch(oracle, 50, sc)
    sc_g, sc_o = dict(sc, U=g1["U"]), dict(sc, U=o1["U"])
    _compare(_gpu_batch(solver, sc_g), _oracle_batch(oracle, 50, sc_o), 1e-8, "tick 2")


def test_committed_regression_vectors(cilqr, solver):
    for c in load_golden("oracle_solves.json")["cases"]:
        N, M, B = c["N"], c["M"], c["B"]
        inp = {k: (None if v is None else np.array(v)) for k, v in c["inputs"].items()}
        got = solver.solve_batch(N, inp["x0"], inp["U"], inp["poly"], inp["xplan_fl"], inp["obs_pose"], inp["obs_dim"])
        want = dict(U=np.array(c["U"]), X=np.array(c["X"]), J=np.array(c["J"]), iters=np.array(c["iters"]), status=np.array(c["status"]))
        _compare(got, want, TIGHT, c["name"])


def test_empty_batch_and_bounds(cilqr, solver):
    import ctypes as C
    L = cilqr.lib()
    z = np.zeros(8)
    zp = z.ctypes.data_as(C.POINTER(C.c_double))
    assert L.cilqr_solve_batch(solver._h, 0, 50, 0, zp, zp, zp, zp, None, None, None, zp, None, None, None, 0) == 0
    assert L.cilqr_solve_batch(solver._h, 4096, 50, 0, zp, zp, zp, zp, None, None, None, zp, None, None, None, 0) == -1
    assert L.cilqr_solve_batch(solver._h, 1, 50, 2, zp, zp, zp, zp, None, None, None, zp, None, None, None, 0) == -1


def test_nonfinite_input_is_contained(cilqr, solver):
    """A NaN start state must not hang or poison neighbours; the solve reports it through status / NaN outputs."""
    from cilqr_amd import scenes
    sc = scenes.make_c2(8, cilqr.default_params(50))
    clean = _gpu_batch(solver, sc)
    sc["x0"] = sc["x0"].copy()
    sc["x0"][3, 1] = np.nan
    r = _gpu_batch(solver, sc)
    keep = np.arange(8) != 3
    assert np.array_equal(r["U"][keep], clean["U"][keep])
    assert r["iters"][3] >= 1


# ------------------------------------------------------------------------------------------------ argmin
def test_argmin_device(cilqr, solver):
    import torch
    J = torch.rand(5000, dtype=torch.float64, device="cuda")
    J[1234] = -1.0
    J[4321] = -1.0  # tie → lowest index
    J[7] = float("nan")
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    solver.argmin_device(torch.cuda.current_stream().cuda_stream, 5000, J.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    assert out.tolist() == [-1.0, 1234.0]


# ------------------------------------------------------------------------------------------------ warp
def test_warp_golden_cases(cilqr, solver):
    for k, c in enumerate(load_golden("ref_gridmap.json")["warp"]):
        sg, dg = cilqr.map_geom(*c["src_geom"]), cilqr.map_geom(*c["dst_geom"])
        src = np.array([np.nan if v is None else v for v in c["src"]], dtype=np.float32).reshape(c["src_shape"], order="F")
        bbox = None if c["bbox"] is None else np.array(c["bbox"], dtype=np.float32).reshape(c["dst_shape"], order="F")
        want = np.array([np.nan if v is None else v for v in c["dst"]], dtype=np.float32).reshape(c["dst_shape"], order="F")
        got, oob = solver.warp_costmap(src, sg, dg, *c["pose"], bbox=bbox)
        assert oob == c["n_out_of_range"], k
        assert np.array_equal(got, want, equal_nan=True), k


def test_warp_config4_frames(cilqr, oracle, solver):
    """BASELINE config 4 at full size (1024² → 1024²): a sample of the 300-frame pose stream + the deliberately
    out-of-range frame, bit-exact against the oracle; plus the size-independent round-trip property below."""
    from cilqr_amd import scenes
    c4 = scenes.make_c4()
    sg, dg = cilqr.map_geom(*c4["src_geom"]), cilqr.map_geom(*c4["dst_geom"])
    osg, odg = oracle.map_geom(*c4["src_geom"]), oracle.map_geom(*c4["dst_geom"])
    poses = [tuple(c4["poses"][k]) for k in (0, 37, 75, 150, 299)] + [(80.0, 80.0, 0.3)]
    for pose in poses:
        got, oob = solver.warp_costmap(c4["src"], sg, dg, *pose)
        want, woob = oracle.warp(c4["src"], osg, odg, *pose, threads=16)
        assert oob == woob
        assert np.array_equal(got, want, equal_nan=True)
    assert oob > 0  # the last frame leaves the source map


def test_warp_identity_and_bbox(cilqr, solver):
    """Size-independent properties: an identity pose with equal geometry copies the map; a bbox layer > 90 overrides."""
    rng = np.random.default_rng(9)
    g = cilqr.map_geom(204.8, 204.8, 0.2, 5.0, -3.0)
    src = np.asfortranarray(rng.integers(0, 101, (g.rows, g.cols)).astype(np.float32))
    # the destination map's position is expressed in the vehicle frame; with V = 0 and theta = 0 frames coincide
    got, oob = solver.warp_costmap(src, g, g, 0.0, 0.0, 0.0)
    assert oob == 0 and np.array_equal(got, src)
    bbox = np.zeros_like(src)
    bbox[100:200, 300:400] = 100.0
    bbox[0, 0] = 90.0  # not > 90: no override
    got, _ = solver.warp_costmap(src, g, g, 0.0, 0.0, 0.0, bbox=bbox)
    assert np.all(got[100:200, 300:400] == 100.0) and got[0, 0] == src[0, 0]
    mask = bbox <= 90
    assert np.array_equal(got[mask], src[mask])


# ------------------------------------------------------------------------------------------------ C++ façade
def test_cpp_adapter_replays_reference_call_sequence(cilqr, oracle, tmp_path):
    """host/ilqr_adapter.{h,cpp}: the reference node's own call sequence (set_global_plan → set_Obstacle →
    clear_uncertainty_map → run_step, twice) in C++, through the C-ABI, on the SURVEY §8(c) scene."""
    import json
    import os
    import subprocess
    from conftest import PKG, ROOT
    from cilqr_amd import scenes
    exe = str(tmp_path / "adapter_replay")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "adapter_replay.cpp"), "-L" + os.path.join(PKG, "lib"), "-lcilqr_hip",
                    "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True)
    case = [c for c in load_golden("survey_known_answers.json")["cases"] if c["N"] == 50 and c["M"] == 4][0]
    out = json.loads(subprocess.run([exe, "50", "4"], check=True, capture_output=True, text=True).stdout)
    t0, t1 = out["ticks"]
    assert t0["iterations"] == case["iterations"] and t0["exit"] == 1
    assert np.max(np.abs(np.array(t0["U"][:2]) - np.array(case["U0"]))) < TIGHT
    assert np.max(np.abs(np.array(t0["X"][-4:]) - np.array(case["XN"]))) < 1e-8
    # second tick: warm start = first tick's U_result, same ego state (I/iLQR.cpp:253) — against the oracle
    po = oracle.default_params(50)
    sc = scenes.known_answer_scene(50, 4, po, local_plan=oracle.local_plan)
    o0 = oracle.solve(po, 50, sc["x0"][0], sc["U"][0], sc["poly"][0], sc["xplan_fl"][0, 0], sc["xplan_fl"][0, 1], sc["obs_pose"][0], sc["obs_dim"][0])
    o1 = oracle.solve(po, 50, sc["x0"][0], o0["U"], sc["poly"][0], sc["xplan_fl"][0, 0], sc["xplan_fl"][0, 1], sc["obs_pose"][0], sc["obs_dim"][0])
    assert t1["iterations"] == o1["iters"] and t1["exit"] == o1["status"]
    assert np.max(np.abs(np.array(t1["U"]) - o1["U"])) < 1e-8
    assert t0["n_ref"] == 20
    # run_candidates: eight perturbed ego states from the warm start left by tick 2, one launch, minimum-cost pick — against
    # the oracle's eight pre-steps + solves and its strict-< first minimum
    i = np.arange(200.0)
    path = np.stack([i, 0.5 * np.sin(0.05 * i)], axis=1)
    Js = []
    for c in range(8):
        ego = np.array([0.05 * c, 0.1 - 0.04 * c, 3.0, 0.02 + 0.01 * c])
        coeffs, ref = oracle.local_plan(po, path, ego)
        r = oracle.solve(po, 50, ego, o1["U"], coeffs, ref[0, 0], ref[-1, 0], sc["obs_pose"][0], sc["obs_dim"][0])
        Js.append(r["J"])
    assert out["best"] == int(np.argmin(Js))
    assert abs(out["best_J"] - min(Js)) < 1e-9 * max(1.0, abs(min(Js)))


# ------------------------------------------------------------------------------------------------ rare branches
def test_general_kernel_paths(cilqr, oracle, solver):
    """Inputs that FORCE the hand-over from the fast kernel to the GENERAL kernel (rule: a rare data-dependent branch needs
    its own test): (a) a heading beyond the in-loop sincos range, (b) negative obstacle weights that make Q_uu indefinite so
    that the eigenvalue clamp of I/iLQR.cpp:167 acts.  Neighbouring ordinary solves in the same batch must be unaffected."""
    from cilqr_amd import scenes
    p = cilqr.default_params(50)
    sc = scenes.make_c2(16, p)
    base = _gpu_batch(solver, sc)
    # (a) huge heading (an exact multiple of 2π keeps the scene geometry; the library-range sincos must be used)
    sa = dict(sc, x0=sc["x0"].copy())
    sa["x0"][5, 3] += 2 * np.pi * 400000  # ≈ 2.5e6 rad
    got, want = _gpu_batch(solver, sa), _oracle_batch(oracle, 50, sa)
    _compare(got, want, 1e-7, "huge heading")  # |theta| ~ 2.5e6: ulp(theta) ~ 5e-10 limits agreement
    keep = np.arange(16) != 5
    assert np.array_equal(got["U"][keep], base["U"][keep])
    # (b) strongly negative obstacle weights make the problem non-convex and numerically chaotic (the reference itself ends
    # such solves on NaN): no value parity is claimed there, only containment — the solve terminates with a legal status
    # and its neighbours are untouched.
    sb = dict(sc, obs_weight=np.full((16, 4), 1.0))
    sb["obs_weight"][3, :] = -40.0
    got = _gpu_batch(solver, sb)
    assert got["status"][3] in (0, 1, 2, 3) and 1 <= got["iters"][3] <= 20
    keep = np.arange(16) != 3
    assert np.array_equal(got["U"][keep], base["U"][keep])


@pytest.mark.parametrize("G", [0, 8])
def test_large_turns_hand_over_to_general_kernel(cilqr, oracle, G, monkeypatch):
    """The production kernels advance the heading's cos/sin by rotation, valid for turns up to 1/4 rad per step
    (rotate_heading, cilqr_device.hpp); a solve that turns faster anywhere must be redone by the GENERAL kernel.  Fast egos
    (10-25 m/s) with a full-lock warm start turn 0.3-0.8 rad per step: they must still follow the oracle, in both families,
    and the ordinary solves beside them in the batch keep their bits; a pass-count buffer shows who handed over."""
    import torch
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 50, 2, 48
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 8111)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        base = _gpu_batch(s, sc)
        fast = dict(sc, x0=sc["x0"].copy(), U=sc["U"].copy())
        rng = np.random.default_rng(8112)
        hot = np.arange(0, B, 3)
        fast["x0"][hot, 2] = rng.uniform(10.0, 25.0, hot.size)
        Uh = fast["U"].reshape(B, N, 2)
        Uh[hot, :, 1] = np.where(rng.random((hot.size, 1)) < 0.5, 8.0, -8.0)  # clamped to ±v·tan(steer_max)/L inside the model
        got = _gpu_batch(s, fast)
    finally:
        s.close()
    want = _oracle_batch(oracle, N, fast)
    ok = np.isfinite(want["U"]).all(axis=1)
    assert ok[hot].sum() >= hot.size // 2
    _compare({k: v[ok] for k, v in got.items()}, {k: v[ok] for k, v in want.items()}, 1e-8, "large turns G=%d" % G)
    assert np.array_equal(got["status"][~ok], want["status"][~ok]) and np.array_equal(got["iters"][~ok], want["iters"][~ok])
    cold = np.setdiff1d(np.arange(B), hot)
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(got[k][cold], base[k][cold]), k
    # the turn really exceeded the bound on the hot solves' first rollout
    yaw_hi = np.tan(p.steer_angle_max) / p.wheelbase
    assert (fast["x0"][hot, 2] * yaw_hi * p.timestep > 0.25).all()


def test_quu_inverse_branches_vs_reference_eigensolver(cilqr, solver):
    """The kernels' regularised Q_uu inverse against the reference's own EigenSolver path (tests/golden/ref_quu.json,
    generated from the vendored Eigen): the eigenvalue-CLAMPING branch of the GENERAL kernel on the indefinite matrices, and
    the production kernel's PSD form on the positive-definite ones."""
    cases = load_golden("ref_quu.json")["cases"]
    # the kernels carry Q_uu as a symmetric matrix: keep the cases that are symmetric to rounding, as every
    # Q_uu the solver forms is; the deliberately non-symmetric fixtures pin the oracle's EigenSolver restatement only
    cases = [c for c in cases if abs(c["Quu"][1] - c["Quu"][2]) <= 1e-14 * max(abs(v) for v in c["Quu"])]
    # ... and drop the exactly degenerate one (a == d, |b| below the solver's deflation threshold): there the reference's
    # EigenSolver returns NON-orthogonal eigenvectors, so its V·D·Vᵀ is off from the true inverse by 0.75 % — reproduced
    # by the oracle (tests/test_oracle.py), not by the kernels, and unreachable in a solve (l_uu has 2·w_acc ≠ 2·w_yawrate)
    cases = [c for c in cases if not (c["Quu"][0] == c["Quu"][3] and abs(c["Quu"][1]) < 1e-12 * abs(c["Quu"][0]) and c["Quu"][1] != 0.0)]
    Q = np.array([c["Quu"] for c in cases])
    lamb = np.array([c["lamb"] for c in cases])
    want = np.array([c["Qinv"] for c in cases])
    ev = np.array([c["eval"] for c in cases])
    scale = np.max(np.abs(want), axis=1)
    gen = solver.debug_quu_inverse(Q, lamb, general=True)
    err = np.max(np.abs(gen - want), axis=1) / scale
    indefinite = ev.min(axis=1) < 0
    assert indefinite.sum() >= 20          # the clamp really is exercised
    assert err.max() < 1e-12, err.max()
    psd = ev.min(axis=1) > 0
    fast = solver.debug_quu_inverse(Q[psd], lamb[psd], general=False)
    cond = ev[psd].max(axis=1) / ev[psd].min(axis=1)
    assert np.max(np.max(np.abs(fast - want[psd]), axis=1) / scale[psd] / np.maximum(cond, 1.0)) < 1e-14
    nan = solver.debug_quu_inverse(np.array([[1.0, np.nan, np.nan, 2.0], [np.inf, 0.0, 0.0, -np.inf]]), np.array([1.0, 1.0]), general=True)
    assert np.isnan(nan).all()


# ------------------------------------------------------------------------------------------------ grouped kernel family
@pytest.mark.parametrize("G", [1, 2, 4, 8, 16, 32])
def test_grouped_kernels_match_oracle(cilqr, oracle, G):
    """The G-lanes-per-solve kernel family (large batches; chosen automatically above one solve per SIMD, B > 1024) forced onto small batches
    through the CILQR_FORCE_G test hook: same parity bar as the wavefront-per-solve family, ragged batch sizes included."""
    import os
    from cilqr_amd import scenes
    os.environ["CILQR_FORCE_G"] = str(G)
    try:
        s = cilqr.Solver(cilqr.default_params(), max_batch=300, max_horizon=80, max_obstacles=16, device=0)
    finally:
        del os.environ["CILQR_FORCE_G"]
    try:
        for N, M, B, seed in ((50, 4, 203, 301), (80, 16, 37, 302), (50, 0, 65, 303), (7, 2, 5, 304), (1, 1, 9, 305), (2, 0, 3, 306),
                              (3, 2, 4, 307)):
            sc = scenes.make_static(B, N, M, cilqr.default_params(N), seed)
            _compare(_gpu_batch(s, sc), _oracle_batch(oracle, N, sc), TIGHT, "G%d N%d M%d" % (G, N, M))
        # hand-over to the GENERAL instantiation (huge heading) and the NaN containment, as for the wavefront family
        sc = scenes.make_c2(16, cilqr.default_params(50))
        base = _gpu_batch(s, sc)
        sa = dict(sc, x0=sc["x0"].copy())
        sa["x0"][5, 3] += 2 * np.pi * 400000
        _compare(_gpu_batch(s, sa), _oracle_batch(oracle, 50, sa), 1e-7, "G%d huge heading" % G)
        sn = dict(sc, x0=sc["x0"].copy())
        sn["x0"][3, 1] = np.nan
        r = _gpu_batch(s, sn)
        keep = np.arange(16) != 3
        assert np.array_equal(r["U"][keep], base["U"][keep])
        # sampled, weighted obstacles (config-3 shape)
        sc3 = scenes.make_c3(24, cilqr.default_params(50), n_dyn=4, n_samples=4)
        _compare(_gpu_batch(s, sc3), _oracle_batch(oracle, 50, sc3), TIGHT, "G%d C3" % G)
        # more obstacles than the 64-bit held-row mask covers (the 6 beyond it are streamed), all of them constant over the horizon
        s.close()
        os.environ["CILQR_FORCE_G"] = str(G)
        try:
            s = cilqr.Solver(cilqr.default_params(), max_batch=16, max_horizon=150, max_obstacles=70, device=0)
        finally:
            del os.environ["CILQR_FORCE_G"]
        sc70 = scenes.make_static(7, 20, 70, cilqr.default_params(20), 308)
        _compare(_gpu_batch(s, sc70), _oracle_batch(oracle, 20, sc70), TIGHT, "G%d M70" % G)
        # a horizon of many staging chunks, not a multiple of the chunk length
        sc150 = scenes.make_static(6, 150, 2, cilqr.default_params(150), 309)
        _compare(_gpu_batch(s, sc150), _oracle_batch(oracle, 150, sc150), 1e-8, "G%d N150" % G)
    finally:
        s.close()


def test_automatic_family_choice_large_batch(cilqr, oracle):
    """B = 9000 > 8 solves per SIMD takes the grouped family automatically (G = 4); a 256-solve sample is checked against the oracle."""
    from cilqr_amd import scenes
    p = cilqr.default_params(50)
    sc = scenes.make_static(9000, 50, 4, p, 401)
    s = cilqr.Solver(p, max_batch=9000, max_horizon=50, max_obstacles=4, device=0)
    try:
        got = _gpu_batch(s, sc)
    finally:
        s.close()
    sub = {k: (v[:256] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    want = _oracle_batch(oracle, 50, sub)
    _compare({k: v[:256] for k, v in got.items()}, want, TIGHT, "auto G")
    assert np.isfinite(got["U"]).all()


@pytest.mark.parametrize("B,N,M", [(2048, 50, 4), (8192, 50, 4), (2048, 80, 16)])
def test_several_wavefronts_per_simd(cilqr, oracle, monkeypatch, B, N, M):
    """Batches of two to eight solves per SIMD on the one-wavefront-per-solve family (forced here: the library keeps it for a
    few solves per SIMD only while the solves are short, pick_group_lanes); N = 80 / M = 16 is the instantiation with the obstacle
    table in global memory.  First, middle and last 64 solves against the oracle, the rest finite with a sane status."""
    from cilqr_amd import scenes
    monkeypatch.setenv("CILQR_FORCE_G", "64")
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 402)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        got = _gpu_batch(s, sc)
    finally:
        s.close()
    idx = np.concatenate([np.arange(64), np.arange(1000, 1064), np.arange(B - 64, B)])
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    want = _oracle_batch(oracle, N, sub)
    _compare({k: v[idx] for k, v in got.items()}, want, TIGHT, "several per SIMD")
    assert np.isfinite(got["U"]).all() and (got["iters"] >= 1).all()


def _pair_vs_single(cilqr, monkeypatch, sc, N, M, B):
    """The same batch on a handle created with CILQR_PAIR_KERNEL (the two-wavefront kernel, an opt-in experiment) and on an ordinary one."""
    p = cilqr.default_params(N)
    monkeypatch.setenv("CILQR_PAIR_KERNEL", "1")
    two = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.delenv("CILQR_PAIR_KERNEL")
    monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")  # (the ordinary handle would give small batches a second wavefront of another kind)
    one = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.delenv("CILQR_NO_SHARE_KERNEL")
    try:
        return _gpu_batch(two, sc), _gpu_batch(one, sc)
    finally:
        two.close()
        one.close()


@pytest.mark.parametrize("N,M,B", [(50, 4, 1024), (30, 2, 200), (2, 1, 9), (3, 0, 5), (17, 5, 64), (64, 4, 96), (72, 3, 40), (33, 9, 70)])
def test_pair_kernel_equals_single_wavefront_kernel(cilqr, oracle, monkeypatch, N, M, B):
    """CILQR_PAIR_KERNEL (opt-in experiment): up to one solve per SIMD every solve runs as a workgroup of two
    wavefronts — the linearisation of the new trajectory on the second one, behind the forward pass, four lanes per step.  Same
    statements, other order of a few sums: it must take the accept / reject path of the one-wavefront kernel on every solve and
    agree with it to rounding (1e-11) — on config 2 in full and on ragged shapes: horizons that are no multiple of its chunks,
    obstacle counts that are no multiple of the quad, none at all, N > 64 — and both must agree with the oracle."""
    from cilqr_amd import scenes
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 7100 + N)
    got, ref = _pair_vs_single(cilqr, monkeypatch, sc, N, M, B)
    _compare(got, ref, 1e-11, "two wavefronts against one")
    idx = np.arange(min(B, 128))
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "two wavefronts per solve")


def test_pair_kernel_weights_warm_starts_and_hand_over(cilqr, oracle, monkeypatch):
    """The two-wavefront kernel with per-obstacle weights, warm-started (random) controls, moving obstacles, and solves that it
    hands to the GENERAL kernel (a NaN start, a heading beyond the in-loop sincos range, a turn of more than 1/4 rad per step):
    the accept / reject paths of the one-wavefront kernel, and the oracle's results where it has finite ones."""
    from cilqr_amd import scenes
    N, M, B = 50, 6, 160
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 7177)
    rng = np.random.default_rng(7177)
    sc["obs_weight"] = rng.uniform(0.2, 2.0, (B, M))
    sc["U"] = sc["U"] + rng.normal(0.0, 0.3, sc["U"].shape)
    pose = sc["obs_pose"].reshape(B, M, N, 4).copy()
    pose[:, 0, :, 2] = 3.0  # a moving obstacle: speed inflates its ellipse (I/Obstacle.cpp:42-43)
    pose[:, 0, :, 0] += 0.3 * np.arange(N)
    sc["obs_pose"] = pose.reshape(B, M, 4 * N)
    sc["x0"][3, 1] = np.nan
    sc["x0"][5, 3] = 2.0e6
    sc["x0"][7, 2] = 25.0
    sc["U"][7, 1::2] = 5.0  # full lock at 25 m/s: more than 1/4 rad per step
    got, ref = _pair_vs_single(cilqr, monkeypatch, sc, N, M, B)
    assert np.array_equal(got["iters"], ref["iters"]) and np.array_equal(got["status"], ref["status"])
    assert np.array_equal(np.isnan(got["U"]), np.isnan(ref["U"]))
    keep = np.ones(B, bool)
    keep[[3]] = False
    sub = {k: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[keep] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "two wavefronts, weights and warm starts")


def _share_vs_single(cilqr, monkeypatch, sc, N, M, B):
    """The same batch on an ordinary handle (cilqr_solve_share_kernel where it applies) and on one created with CILQR_NO_SHARE_KERNEL."""
    p = cilqr.default_params(N)
    two = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")
    one = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.delenv("CILQR_NO_SHARE_KERNEL")
    try:
        return _gpu_batch(two, sc), _gpu_batch(one, sc), two.solve_wavefronts(B, N, M), one.solve_wavefronts(B, N, M)
    finally:
        two.close()
        one.close()


def _same_bits(got, ref, what):
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(got[k], ref[k], equal_nan=(k in ("U", "X", "J"))), "%s: %s differs" % (what, k)


@pytest.mark.parametrize("N,M,B", [(50, 4, 1024), (50, 4, 768), (50, 4, 2048), (50, 12, 300), (50, 8, 1024), (50, 30, 64), (50, 20, 400), (80, 16, 64), (64, 3, 100), (64, 1, 40), (65, 2, 50), (127, 2, 30), (100, 6, 300), (30, 2, 200), (2, 1, 9), (1, 1, 3), (3, 0, 5), (17, 5, 64), (63, 4, 96), (33, 9, 70), (40, 3, 33)])
def test_share_kernel_changes_no_bit(cilqr, oracle, monkeypatch, N, M, B):
    """Up to two solves per SIMD a static-obstacle solve runs as a workgroup of two or three wavefronts that work on phase L at the same
    time (cilqr_solve_share_kernel: closest samples and tracking terms on one; cos / sin, obstacle sums — with three wavefronts the
    entries of even index on one, of odd index on the other —, Jacobians and control barrier on the others).  The statements are
    lin_step's, whose obstacle terms are summed from zero in an even and an odd chain in every kernel: U, X, J, iterations and exit
    reasons must be BIT-IDENTICAL to the one-wavefront kernel's — config 2 in full, two solves per SIMD, ragged shapes (N = 1, 2, 63;
    no obstacles, one, odd counts), crowded scenes whose table only fits LDS because few solves share a CU, horizons of 64 … 127 (two
    steps per lane) — with three wavefronts and with two, and agree with the oracle."""
    from cilqr_amd import scenes
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 7300 + N)
    got, ref, w2, w1 = _share_vs_single(cilqr, monkeypatch, sc, N, M, B)
    assert (w2, w1) == (3 if B <= 768 and M >= 2 and N <= 64 else 2, 1)  # (MI355X: 1024 SIMDs; three wavefronts up to 3/4 solve per SIMD, N ≤ 64)
    _same_bits(got, ref, "%d wavefronts sharing phase L against one" % w2)
    if w2 == 3:  # the other wavefront count on the same shape
        monkeypatch.setenv("CILQR_SHARE_W", "2")
        got2, _, w, _ = _share_vs_single(cilqr, monkeypatch, sc, N, M, B)
        monkeypatch.delenv("CILQR_SHARE_W")
        assert w == 2
        _same_bits(got2, ref, "two wavefronts sharing phase L against one")
    idx = np.arange(min(B, 128))
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "two wavefronts sharing phase L")


@pytest.mark.parametrize("M", [6, 30])
def test_share_kernel_weights_warm_starts_and_hand_over(cilqr, oracle, monkeypatch, M):
    """The shared-phase-L kernel with per-obstacle weights, warm-started (random) controls, a moving obstacle, and solves that it hands
    to the GENERAL kernel (a NaN start, a heading beyond the in-loop sincos range, a turn of more than 1/4 rad per step): the bits of
    the one-wavefront kernel, NaNs included, and the oracle's results where it has finite ones.  Where the kernel does not apply
    (N = 128: two steps per lane no longer cover the horizon; a batch beyond two solves per SIMD) the library says so and runs one wavefront."""
    from cilqr_amd import scenes
    N, B = 50, 160  # (M = 30: 87 KB of LDS per solve — the table only fits because one solve has a CU to itself; both kernels of the pair need their limit raised)
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 7377)
    rng = np.random.default_rng(7377)
    sc["obs_weight"] = rng.uniform(0.2, 2.0, (B, M))
    sc["U"] = sc["U"] + rng.normal(0.0, 0.3, sc["U"].shape)
    pose = sc["obs_pose"].reshape(B, M, N, 4).copy()
    pose[:, 0, :, 2] = 3.0  # a moving obstacle: speed inflates its ellipse (I/Obstacle.cpp:42-43)
    pose[:, 0, :, 0] += 0.3 * np.arange(N)
    sc["obs_pose"] = pose.reshape(B, M, 4 * N)
    sc["x0"][3, 1] = np.nan
    sc["x0"][5, 3] = 2.0e6
    sc["x0"][7, 2] = 25.0
    sc["U"][7, 1::2] = 5.0  # full lock at 25 m/s: more than 1/4 rad per step
    got, ref, w2, w1 = _share_vs_single(cilqr, monkeypatch, sc, N, M, B)
    assert (w2, w1) == (3, 1)
    _same_bits(got, ref, "shared phase L, weights and warm starts")
    keep = np.ones(B, bool)
    keep[[3]] = False
    sub = {k: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[keep] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "shared phase L, weights and warm starts")
    s = cilqr.Solver(cilqr.default_params(64), max_batch=8192, max_horizon=64, max_obstacles=4, device=0)
    try:
        simds = 1024  # MI355X: 256 CUs × 4
        assert s.solve_wavefronts(64, 128, 4) == 1 and s.solve_wavefronts(64, 127, 4) == 2 and s.solve_wavefronts(64, 65, 4) == 2 and s.solve_wavefronts(64, 64, 4) == 3
        assert s.solve_wavefronts(64, 63, 4) == 3 and s.solve_wavefronts(64, 63, 1) == 2
        assert s.solve_wavefronts(3 * simds // 4, 50, 4) == 3 and s.solve_wavefronts(3 * simds // 4 + 1, 50, 4) == 2
        assert s.solve_wavefronts(2 * simds, 50, 4) == 2 and s.solve_wavefronts(2 * simds + 1, 50, 4) == 1
        assert s.solve_wavefronts(64, 50, 80) == 1  # (the table of 80 obstacles × 50 steps does not fit a CU's LDS)
        # a solve's LDS share grows where fewer solves share a CU: twelve obstacles fit up to two solves per CU, not at four; forty at one
        assert s.solve_wavefronts(512, 50, 12) == 3 and s.solve_wavefronts(1024, 50, 12) == 1 and s.solve_wavefronts(1024, 50, 8) == 2
        assert s.solve_wavefronts(64, 50, 40) == 3 and s.solve_wavefronts(300, 50, 40) == 1
    finally:
        s.close()


@pytest.mark.parametrize("G,B,N,M", [(8, 520, 80, 16), (1, 300, 30, 2), (4, 333, 50, 4), (16, 90, 50, 5), (32, 40, 64, 3), (2, 257, 20, 0)])
def test_lane_sharing_changes_no_bit(cilqr, oracle, monkeypatch, G, B, N, M):
    """Grouped family, phase L: the lanes of a wavefront's finished solves take steps of the unfinished ones (floor(64 / k) lanes
    per solve with k solves active).  The result of a solve must not depend on it: bit-identical U, X, J, iterations and exits
    with the sharing switched off (CILQR_NO_LANE_SHARING), for every lane grouping, ragged batches (the last wavefront partly
    empty), warm-started controls and per-obstacle weights; and equal to the oracle."""
    from cilqr_amd import scenes
    monkeypatch.setenv("CILQR_FORCE_G", str(G))
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 7300 + G)
    rng = np.random.default_rng(7300 + G)
    sc["U"] = sc["U"] + rng.normal(0.0, 0.2, sc["U"].shape)
    if M:
        sc["obs_weight"] = rng.uniform(0.3, 1.5, (B, M))
    shared = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.setenv("CILQR_NO_LANE_SHARING", "1")
    plain = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    monkeypatch.delenv("CILQR_NO_LANE_SHARING")
    try:
        assert shared.solve_family(B, N, M) == G
        got, ref = _gpu_batch(shared, sc), _gpu_batch(plain, sc)
        flags = cilqr.FLAG_FAITHFUL_ITERS
        gotf, reff = _gpu_batch(shared, sc, flags=flags), _gpu_batch(plain, sc, flags=flags)
    finally:
        shared.close()
        plain.close()
    for k in ("iters", "status", "U", "X", "J"):
        assert np.array_equal(got[k], ref[k]), k
        assert np.array_equal(gotf[k], reff[k]), "faithful " + k
        assert np.array_equal(got[k], gotf[k]), "early exit against the reference loop: " + k
    idx = np.arange(min(B, 96))
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "lane sharing, G=%d" % G)


def test_family_rule_on_measured_shapes(cilqr, oracle):
    """The family rule (pick_group_lanes, drawn from profiles/r03_family_shapes.txt) at shapes on either side of its lines, and
    one long-horizon batch end to end: N = 120, B = 4096 takes the wavefront family (5.9 ms against 6.7 ms grouped) and agrees with
    the oracle on a sample."""
    from cilqr_amd import scenes
    N, M, B = 120, 4, 4096
    p = cilqr.default_params(N)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=16, device=0)
    try:
        fam = s.solve_family
        assert fam(1024, 50, 4) == 64 and fam(4096, 50, 4) == 64 and fam(8192, 50, 4) == 8 and fam(4096, 50, 8) == 64 and fam(8192, 50, 8) == 8
        assert fam(8192, 30, 2) == 64 and fam(16384, 30, 2) == 4 and fam(2048, 64, 4) == 64 and fam(4096, 64, 4) == 16
        assert fam(2048, 80, 16) == 64 and fam(4096, 80, 16) == 16 and fam(8192, 80, 16) == 8 and fam(65536, 80, 16) == 4 and fam(65536, 50, 4) == 2
        assert fam(4096, 120, 4) == 64 and fam(16384, 160, 16) == 64 and fam(4096, 50, 256) == 64
        sc = scenes.make_static(B, N, M, p, 7400)
        got = _gpu_batch(s, sc)
    finally:
        s.close()
    idx = np.concatenate([np.arange(32), np.arange(2000, 2032), np.arange(B - 32, B)])
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "N = 120, B = 4096")
    assert (got["iters"] >= 1).all()


@pytest.mark.parametrize("B,N,n_dyn,S,W", [(200, 50, 8, 32, 4), (1200, 50, 8, 8, 2), (33, 30, 3, 5, 2), (40, 64, 2, 4, 2), (17, 12, 5, 2, 4), (9, 20, 7, 3, 4)])
def test_split_kernel_against_one_wavefront_per_solve(cilqr, oracle, monkeypatch, B, N, n_dyn, S, W):
    """Sampled obstacles: W = 2 or 4 wavefronts per solve share the obstacle entries of phase L (cilqr_solve_split_kernel; the default
    for horizons up to 64: four up to one solve per SIMD, two beyond).  Against the one-wavefront kernel (CILQR_NO_SPLIT_KERNEL): the same accept / reject path on every solve and
    agreement to rounding (the entries of a step are summed in two interleaved halves); against the oracle on the materialised
    scene: TIGHT.  Odd obstacle counts (the halves differ in size), horizons of 12 … 64, obstacles that turn and brake."""
    from cilqr_amd import scenes
    p = cilqr.default_params(N)
    sc = scenes.make_c3(B, p, n_dyn=n_dyn, n_samples=S) if N == 50 else None
    if sc is None:  # other horizons: the static generator's scene with moving, turning obstacles and drawn offsets
        st = scenes.make_static(B, N, n_dyn, p, 7500 + N)
        rng = np.random.default_rng(7500 + N)
        pose = st["obs_pose"].reshape(B, n_dyn, N, 4).copy()
        pose[..., 2] = rng.uniform(0.0, 6.0, (B, n_dyn, 1))
        pose[:, 0, :, 3] += 0.01 * np.arange(N)  # one obstacle turns: the per-entry derivation of its samples
        off = rng.normal(0.0, 1.0, (B, n_dyn, S, 3)) * scenes.POSE_SIGMA
        mp, md, mw = scenes.materialise_samples(pose.reshape(B, n_dyn, 4 * N), st["obs_dim"], off, N)
        sc = dict(st, M=n_dyn * S, nom_pose=pose.reshape(B, n_dyn, 4 * N), nom_dim=st["obs_dim"], offsets=off, sample_weight=1.0 / S,
                  obs_pose=mp, obs_dim=md, obs_weight=mw)

    def run(slv):
        return slv.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"], sc["offsets"], sc["sample_weight"])
    monkeypatch.setenv("CILQR_SPLIT_W", str(W))
    two = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=n_dyn * S, device=0)
    monkeypatch.delenv("CILQR_SPLIT_W")
    monkeypatch.setenv("CILQR_NO_SPLIT_KERNEL", "1")
    one = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=n_dyn * S, device=0)
    monkeypatch.delenv("CILQR_NO_SPLIT_KERNEL")
    try:
        got, ref, again = run(two), run(one), run(two)
    finally:
        two.close()
        one.close()
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(got[k], again[k]), k
    _compare(got, ref, 1e-11, "split kernel against one wavefront per solve")
    idx = np.arange(min(B, 48))
    sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "split kernel")


@pytest.mark.parametrize("B,W", [(160, 4), (300, 2)])
def test_split_kernel_with_uncertainty_map(cilqr, oracle, monkeypatch, B, W):
    """Sampled obstacles AND an uncertainty map — the reference planner's full mode: the split kernel gives the map term to its last
    wavefront.  Against the one-wavefront kernel with the same map (CILQR_NO_SPLIT_KERNEL): same accept / reject paths, agreement to
    rounding; against the oracle on the materialised scene with the same map: 1e-8."""
    from cilqr_amd import scenes
    N = 50
    p, po = _unc_params(cilqr, N), _unc_params(oracle, N)
    sc = scenes.make_c3(B, p, n_dyn=5, n_samples=6)
    geom, layer = _unc_layer(oracle, 3)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    pose = (-0.5, 0.3, 0.04)
    out = {}
    for name in ("split", "one"):
        monkeypatch.setenv("CILQR_SPLIT_W", str(W)) if name == "split" else monkeypatch.setenv("CILQR_NO_SPLIT_KERNEL", "1")
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=sc["M"], device=0)
        monkeypatch.delenv("CILQR_SPLIT_W", raising=False)
        monkeypatch.delenv("CILQR_NO_SPLIT_KERNEL", raising=False)
        try:
            s.set_uncertainty_map(layer, g, pose, (3, 3))
            w = s.solve_sampled_wavefronts(B, N, 5)
            out[name] = (w, s.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"], sc["offsets"],
                                                  sc["sample_weight"]))
        finally:
            s.close()
    assert (out["split"][0], out["one"][0]) == (W, 1)
    _compare(out["split"][1], out["one"][1], 1e-11, "split kernel with a map against one wavefront per solve")
    um, keep = oracle.uncertainty_map(layer, og, pose, (3, 3))
    want = oracle.solve_batch_unc(po, N, sc["M"], sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"], um,
                                  threads=min(16, oracle.max_threads()))
    _compare(out["split"][1], want, 1e-8, "split kernel with a map")


@pytest.mark.parametrize("B", [3000, 1800])
def test_schedule_hint_changes_nothing_but_the_order(cilqr, B):
    """A batch beyond one solve per SIMD is dispatched longest-first by the pass counts of the previous call (same batch size,
    same stream).  Every call must return bit-identical results — the first (identity order), the second (hinted) and a third
    after the scenes were shuffled, when the hint is stale — and every solve must be written exactly once.  B = 3000: one wavefront
    per solve; B = 1800: two (cilqr_solve_share_kernel takes the dispatch order too)."""
    from cilqr_amd import scenes
    N, M = 50, 4
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 403)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        assert s.solve_wavefronts(B, N, M) == (2 if B == 1800 else 1)
        first = _gpu_batch(s, sc)
        second = _gpu_batch(s, sc)
        perm = np.random.default_rng(5).permutation(B)
        scp = dict(sc)
        for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"):
            scp[k] = np.ascontiguousarray(sc[k][perm])
        third = _gpu_batch(s, scp)
    finally:
        s.close()
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(first[k], second[k]), k
        assert np.array_equal(first[k][perm], third[k]), k
    assert (first["iters"] >= 1).all() and np.isfinite(first["U"]).all()


# ------------------------------------------------------------------------------------------------ uncertainty blur
def _ulp32_diff(a, b):
    """Distance in float32 ulps between same-shaped arrays; NaN pairs count as 0, NaN vs number as huge."""
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    d = np.abs(ai - bi)
    both_nan = np.isnan(a) & np.isnan(b)
    one_nan = np.isnan(a) ^ np.isnan(b)
    return np.where(both_nan, 0, np.where(one_nan, 1 << 40, d))


@pytest.mark.parametrize("geom,sigma,theta,index", [
    ((30.0, 20.0, 0.2, 15.0, 0.0), (0.16, 0.16, 0.017), -1.2, 0),      # the node's default vehicle map, launch-file sigmas
    ((30.0, 20.0, 0.2, 15.0, 0.0), (0.005, 0.005, 0.0125), 0.3, 40),   # dynamic_reconfigure defaults: many NaN-axis cells
    ((30.0, 20.0, 0.2, 10.0, 0.0), (0.3, 0.2, 0.05), 2.5, 0),          # large ellipses (up to ~370 cells)
    ((102.4, 102.4, 0.1, 5.0, -3.0), (0.16, 0.16, 0.017), 0.3, 0),     # 1024 × 1024 cells (one lane per cell)
    ((40.0, 30.0, 0.1, 12.0, 1.5), (0.16, 0.16, 0.017), 0.9, 7),       # 400 × 300 cells: the 4-lanes-per-cell instantiation
    ((25.6, 25.6, 0.1, 0.0, 0.0), (0.1, 0.25, 0.03), -2.2, 0),         # 256 × 256 cells: 4 lanes per cell, anisotropic sigmas
])
def test_blur_kernel_vs_oracle(cilqr, oracle, solver, geom, sigma, theta, index):
    """Fused blur kernel against the oracle (itself bit-equal to the reference's grid_map_core + Eigen, tests/test_oracle.py).
    Tolerance: ellipse membership counts EQUAL; outputs within 1 float32 ulp (the density uses hoisted reciprocals:
    ~1e-16 relative in fp64 before the final cast), at least 99.9 % of them bit-equal."""
    rng = np.random.default_rng(31)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    src = rng.integers(0, 101, (g.rows, g.cols)).astype(np.float32)
    src[rng.random(src.shape) < 0.01] = np.nan
    got, cnt = solver.blur_costmap(src, g, theta, *sigma, index=index)
    want, wcnt, _ = oracle.blur(src, og, np.sin(theta), np.cos(theta), *sigma, index=index, threads=16)
    assert np.array_equal(cnt[index:], wcnt[index:])
    d = _ulp32_diff(np.ascontiguousarray(got.flatten(order="F")), np.ascontiguousarray(want.flatten(order="F")))
    assert d.max() <= 1, d.max()
    assert (d == 0).mean() >= 0.999
    assert np.isnan(got.flatten(order="F")[:index]).all()


def test_blur_kernel_vs_reference_golden(cilqr, solver):
    for c in load_golden("ref_blur.json")["cases"]:
        shape = c["shape"]
        src = np.array([np.nan if v is None else v for v in c["src"]], dtype=np.float32).reshape(shape, order="F")
        want = np.array([np.nan if v is None else v for v in c["out"]], dtype=np.float32).reshape(shape, order="F")
        g = cilqr.map_geom(*c["geom"])
        got, cnt = solver.blur_costmap(src, g, c["theta"], *c["sigma"], index=c["index"])
        keep = np.ones(cnt.size, bool)
        keep[c["edge_ub"]] = False
        keep[:c["index"]] = False
        assert np.array_equal(cnt[keep], np.array(c["count"])[keep])
        d = _ulp32_diff(np.ascontiguousarray(got.flatten(order="F")), np.ascontiguousarray(want.flatten(order="F")))
        assert d[keep].max() <= 1


def test_blur_ellipse_step_vs_reference_eigen(cilqr, solver):
    """The kernel's covariance → confidence-ellipse step (float eigen-solve) against the values the reference's own
    Eigen::EigenSolver<Matrix2f> produced (ref_blur.json): half axes bit-equal (NaN where the reference has NaN — slightly
    negative float eigenvalues), angle within 4 double ulps (atan2 of identical float inputs, different libm)."""
    for c in load_golden("ref_blur.json")["cases"]:
        g = cilqr.map_geom(*c["geom"])
        lin = np.arange(c["index"], g.rows * g.cols)
        ci, cj = lin % g.rows, lin // g.rows
        Cx = (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * (-ci.astype(float))
        Cy = (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * (-cj.astype(float))
        s, co = np.sin(c["theta"]), np.cos(c["theta"])
        sx, sy, st = c["sigma"]
        u = (-s * Cx - co * Cy) * (-s * Cx - co * Cy)
        v = (co * Cx - s * Cy) * (co * Cx - s * Cy)
        t = s * co * (Cx * Cx - Cy * Cy) + Cx * Cy * (s * s - co * co)
        sxi, syi = np.sqrt(sx * sx + st * st * u), np.sqrt(sy * sy + st * st * v)
        rho = st * st * t / (sxi * syi)
        got = solver.debug_blur_ellipse(np.stack([sxi * sxi, rho * sxi * syi, syi * syi], 1))
        want = np.array([[np.nan if x is None else x for x in c["ellipse"][k]] for k in lin])
        assert np.array_equal(got[:, :2], want[:, :2], equal_nan=True)
        ok = np.isfinite(want[:, 2])
        assert np.max(np.abs(got[ok, 2] - want[ok, 2]) / np.spacing(np.abs(want[ok, 2]))) <= 4
        assert np.isnan(want[:, 1]).sum() == np.isnan(got[:, 1]).sum()


# ---- batched LocalPlanner on the device -------------------------------------------------------------------
def _oracle_plans(O, p, paths, egos):
    polys, fls, ns, refs = [], [], [], []
    for b in range(egos.shape[0]):
        path = paths if paths.ndim == 2 else paths[b]
        c, ref = O.local_plan(p, path, egos[b])
        polys.append(c)
        fls.append([ref[0, 0], ref[-1, 0]])
        ns.append(ref.shape[0])
        refs.append(ref)
    return np.array(polys), np.array(fls), np.array(ns), refs


def test_local_plan_batch_integer_abscissae_bit_exact(cilqr, oracle, solver):
    """Waypoints at integer x (the SURVEY §8c scene): every power is exact, so the device fit must equal the oracle's
    (itself pinned on the reference's Eigen colPivHouseholderQr, tests/golden/ref_polyfit.json) bit for bit —
    including slices cut short by the end of the path (n < 20, and n < 6: more unknowns than rows)."""
    i = np.arange(200.0)
    path = np.stack([i, 0.5 * np.sin(0.05 * i)], axis=1)
    rng = np.random.default_rng(11)
    B = 512
    s = rng.uniform(0, 199, B)
    s[:8] = [0, 199, 198.4, 196.2, 194.4, 181.0, 180.49, 179.5]
    egos = np.stack([s, 0.5 * np.sin(0.05 * s) + rng.uniform(-2, 2, B), rng.uniform(0, 8, B), rng.uniform(-1, 1, B)], axis=1)
    got = solver.local_plan_batch(path, egos)
    p = oracle.default_params(50)
    poly, fl, n, refs = _oracle_plans(oracle, p, path, egos)
    assert np.array_equal(got["n"], n)
    assert set(n[:8]) >= {1, 2, 4, 6, 19, 20}
    assert np.array_equal(got["xplan_fl"], fl)
    assert np.array_equal(got["poly"], poly)
    for b in range(B):
        assert np.array_equal(got["ref_traj"][b, :n[b]], refs[b])


def test_local_plan_batch_general_paths(cilqr, oracle, solver):
    """Arbitrary abscissae, one path per candidate, in global coordinates of a few hundred metres (where the reference's
    fit is rank-deficient and the column pivoting decides what survives).  x^j on the device is the correctly rounded
    power; the oracle calls libm's pow like the reference, and this image's glibc misrounds about one call in a thousand
    (measured against exact rational arithmetic: 11-23 of 20 000 per exponent) — with 80 non-trivial entries per fit that
    is a last-bit difference in one entry of ≈ 6 % of the matrices.  Hence: the same slice always; bit-equal coefficients
    in ≥ 90 % of fits (observed 93 %); and everywhere the same fitted curve over the slice to 1e-9 m."""
    rng = np.random.default_rng(12)
    B, P = 1024, 60
    x = rng.uniform(-300, 300, (B, 1)) + np.cumsum(rng.uniform(0.5, 1.5, (B, P)), axis=1)
    A, w, ph = rng.uniform(0, 1.5, (B, 1)), rng.uniform(0.02, 0.08, (B, 1)), rng.uniform(0, 2 * np.pi, (B, 1))
    paths = np.stack([x, A * np.sin(w * x + ph)], axis=2)
    k = rng.integers(0, P, B)
    egos = np.stack([x[np.arange(B), k] + rng.uniform(-0.4, 0.4, B), paths[np.arange(B), k, 1] + rng.uniform(-2, 2, B),
                     rng.uniform(0, 8, B), rng.uniform(-1, 1, B)], axis=1)
    got = solver.local_plan_batch(paths, egos)
    p = oracle.default_params(50)
    poly, fl, n, refs = _oracle_plans(oracle, p, paths, egos)
    assert np.array_equal(got["n"], n)
    assert np.array_equal(got["xplan_fl"], fl)
    same = np.all(got["poly"] == poly, axis=1)
    assert same.mean() >= 0.90, same.mean()
    worst = max(np.max(np.abs(got["ref_traj"][b, :n[b], 1] - refs[b][:, 1])) for b in range(B))
    assert worst <= 1e-9, worst


def test_plan_then_solve_on_device_matches_oracle(cilqr, oracle, solver):
    """Raw (global_path, ego) → device pre-step → device solve, against the oracle's pre-step + solve."""
    from cilqr_amd import scenes
    N, M, B = 50, 4, 256
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 991)
    i = np.arange(200.0)
    rng = np.random.default_rng(13)
    A, w, ph = rng.uniform(0, 1.5, (B, 1)), rng.uniform(0.02, 0.08, (B, 1)), rng.uniform(0, 2 * np.pi, (B, 1))
    paths = np.stack([np.broadcast_to(i, (B, 200)), A * np.sin(w * i + ph)], axis=2)
    egos = sc["x0"].copy()
    egos[:, 0] = rng.uniform(0, 150, B)
    egos[:, 1] = (A * np.sin(w * egos[:, :1] + ph))[:, 0] + rng.uniform(-0.3, 0.3, B)
    plan = solver.local_plan_batch(paths, egos)
    got = solver.solve_batch(N, egos, sc["U"], plan["poly"], plan["xplan_fl"])
    po = oracle.default_params(N)
    poly, fl, n, _ = _oracle_plans(oracle, po, paths, egos)
    want = oracle.solve_batch(po, N, 0, egos, sc["U"], poly, fl, None, None, None, threads=min(16, oracle.max_threads()))
    _compare(got, want, TIGHT, "plan+solve")


# ---- OccupancyGrid <-> layer and the fused frame ---------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 3, 4, 150 * 100, 1506 * 1506, 1024 * 1024 + 3])
def test_occupancy_conversions_bit_exact(cilqr, oracle, solver, n):
    """Byte work: bit-exact against the oracle, for sizes on and off the 4-cell vector path (and empty)."""
    rng = np.random.default_rng(n + 1)
    occ = rng.integers(-1, 101, n).astype(np.int8)
    layer = solver.occupancy_to_layer(occ)
    want = oracle.occupancy_to_layer(occ)
    assert np.array_equal(layer.view(np.uint32) & 0x7FFFFFFF > 0x7F800000, np.isnan(want))  # NaN where unknown
    assert np.array_equal(layer[~np.isnan(want)], want[~np.isnan(want)])
    # reference round trip (GridMapRosTest.cpp:140-184)
    assert np.array_equal(solver.layer_to_occupancy(layer, -1.0, 100.0), occ)
    # arbitrary float payload with the node's range, incl. NaN, ±inf, negatives, > 100 and values a hair below integers
    vals = rng.uniform(-20, 130, n).astype(np.float32)
    if n:
        vals[rng.integers(0, n, max(1, n // 50))] = np.nan
        vals[rng.integers(0, n, max(1, n // 97))] = np.float32(np.inf)
        pick = rng.integers(0, n, max(1, n // 7))
        vals[pick] = np.nextafter(np.round(vals[pick]), np.float32(-1e9)).astype(np.float32)
    for lo, hi in ((0.0, 100.0), (-1.0, 100.0), (0.0, 1.0), (5.0, 5.0)):
        assert np.array_equal(solver.layer_to_occupancy(vals, lo, hi), oracle.layer_to_occupancy(vals, lo, hi)), (lo, hi)


def test_costmap_frame_equals_its_three_steps(cilqr, oracle, solver):
    """cilqr_costmap_frame_device (warp → blur → OccupancyGrid written by the blur kernel) against the oracle's three
    steps on the map node's geometry (150×100 vehicle map at 0.2 m, M/src/local_costmap.cpp:212; launch-file sigmas)."""
    import torch
    rng = np.random.default_rng(77)
    sg = cilqr.map_geom(120.0, 120.0, 0.2, 3.0, -2.0)
    dg = cilqr.map_geom(30.0, 20.0, 0.2, 10.0 - 5, 0.0)
    src = np.zeros((sg.rows, sg.cols), dtype=np.float32, order="F")
    for _ in range(60):
        i, j = rng.integers(0, sg.rows - 30), rng.integers(0, sg.cols - 30)
        src[i:i + rng.integers(3, 30), j:j + rng.integers(3, 30)] = 100.0
    src[rng.random(src.shape) < 0.02] = np.nan
    bbox = np.zeros((dg.rows, dg.cols), dtype=np.float32, order="F")
    bbox[40:60, 30:45] = 100.0
    vx, vy, th = 7.5, -4.25, 0.83
    sx, sy, st = 0.16, 0.16, 0.017
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape(-1, order="F"))).to(dev)  # noqa: E731
    d_src, d_bbox = t(src), t(bbox)
    nd = dg.rows * dg.cols
    d_veh = torch.zeros(nd, dtype=torch.float32, device=dev)
    d_unc = torch.zeros(nd, dtype=torch.float32, device=dev)
    d_occ = torch.zeros(nd, dtype=torch.int8, device=dev)
    d_oob = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    solver.costmap_frame_device(stream, d_src.data_ptr(), sg, dg, vx, vy, th, sx, sy, st, d_veh.data_ptr(), d_unc.data_ptr(),
                                d_occ.data_ptr(), bbox=d_bbox.data_ptr(), n_oob=d_oob.data_ptr())
    torch.cuda.synchronize()
    osg = oracle.map_geom(120.0, 120.0, 0.2, 3.0, -2.0)
    odg = oracle.map_geom(30.0, 20.0, 0.2, 10.0 - 5, 0.0)
    w_veh, n_oob = oracle.warp(src, osg, odg, vx, vy, th, bbox=bbox)
    assert int(d_oob.item()) == n_oob == 0
    veh = d_veh.cpu().numpy()
    assert np.array_equal(np.isnan(veh), np.isnan(w_veh.reshape(-1, order="F")))
    assert np.array_equal(veh[~np.isnan(veh)], w_veh.reshape(-1, order="F")[~np.isnan(veh)])
    # blur of a NaN-free copy as well would hide the NaN handling: keep the NaNs, compare where the oracle is finite
    w_unc, _, _ = oracle.blur(w_veh, odg, np.sin(th), np.cos(th), sx, sy, st, threads=min(16, oracle.max_threads()))
    unc = d_unc.cpu().numpy()
    w_unc = w_unc.reshape(-1, order="F")
    fin = ~np.isnan(w_unc)
    assert np.array_equal(np.isnan(unc), ~fin)
    ulp = np.abs(unc[fin].view(np.int32).astype(np.int64) - w_unc[fin].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() >= 0.999
    # the published grid is the conversion of the kernel's own float layer (bit-exact), and of the oracle's wherever the
    # two float layers agree
    occ = d_occ.cpu().numpy()
    assert np.array_equal(occ, oracle.layer_to_occupancy(unc, 0.0, 100.0))
    same = np.ones(nd, dtype=bool)
    same[fin] = ulp == 0
    assert np.array_equal(occ[::-1][same], oracle.layer_to_occupancy(w_unc, 0.0, 100.0)[::-1][same])


def test_error_behaviour_with_a_live_handle(cilqr):
    """The reference solver never throws (I/iLQR.cpp:240-242 just prints); the C-ABI turns misuse into return codes:
    sizes above what cilqr_create reserved, missing obstacle tables, null required pointers — and an empty batch is OK."""
    from cilqr_amd import scenes
    p = cilqr.default_params(50)
    s = cilqr.Solver(p, max_batch=8, max_horizon=50, max_obstacles=2, device=0)
    try:
        sc = scenes.make_static(8, 50, 2, p, 5)
        ok = s.solve_batch(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"])
        assert ok["iters"].min() >= 1
        big = scenes.make_static(9, 50, 2, p, 5)
        with pytest.raises(cilqr.CilqrError, match="-1"):  # CILQR_ERR_ARG: B > max_batch
            s.solve_batch(50, big["x0"], big["U"], big["poly"], big["xplan_fl"], big["obs_pose"], big["obs_dim"])
        m3 = scenes.make_static(8, 50, 3, p, 5)
        with pytest.raises(cilqr.CilqrError, match="-1"):  # M > max_obstacles
            s.solve_batch(50, m3["x0"], m3["U"], m3["poly"], m3["xplan_fl"], m3["obs_pose"], m3["obs_dim"])
        n60 = scenes.make_static(8, 60, 2, cilqr.default_params(60), 5)
        with pytest.raises(cilqr.CilqrError, match="-1"):  # N > max_horizon
            s.solve_batch(60, n60["x0"], n60["U"], n60["poly"], n60["xplan_fl"], n60["obs_pose"], n60["obs_dim"])
        L = cilqr.lib()
        import ctypes as C
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        X = np.zeros((8, 4 * 51))
        U = sc["U"].copy()
        # M > 0 with null obstacle tables; null X_out
        assert L.cilqr_solve_batch(s._h, 8, 50, 2, dp(sc["x0"]), dp(U), dp(sc["poly"]), dp(sc["xplan_fl"]), None, None, None,
                                   dp(X), None, None, None, C.c_uint32(0)) == -1
        assert b"obstacle" in L.cilqr_last_error()
        assert L.cilqr_solve_batch(s._h, 8, 50, 0, dp(sc["x0"]), dp(U), dp(sc["poly"]), dp(sc["xplan_fl"]), None, None, None,
                                   None, None, None, None, C.c_uint32(0)) == -1
        # empty batch: nothing to do, not an error
        assert L.cilqr_solve_batch(s._h, 0, 50, 0, None, None, None, None, None, None, None, None, None, None, None, C.c_uint32(0)) == 0
        # the handle still works after the refusals
        again = s.solve_batch(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"])
        assert np.array_equal(again["U"], ok["U"])
    finally:
        s.close()
    q = cilqr.default_params(50)
    q.num_states = 5  # BASELINE.json says nx=5; the reference is nx=4 and the library refuses to guess
    with pytest.raises(cilqr.CilqrError, match="-4"):
        cilqr.Solver(q, max_batch=1, max_horizon=50, max_obstacles=0, device=0)


def test_replay_tool_closed_loop_against_oracle(cilqr, oracle, tmp_path):
    """bin/cilqr_replay (host/replay_main.cpp): six recorded ticks — ego advancing along the previous plan, an obstacle
    appearing at tick 2 and gone at tick 5, the warm start carried un-shifted from tick to tick (I/iLQR.cpp:253) — replayed
    through the adapter and compared, tick by tick, with the oracle driven the same way.  Output is the
    vehiclepub/Experiment flattening (I/ilqr_uncertainty_node.cpp:243-284)."""
    import os
    import subprocess
    from conftest import PKG
    exe = os.path.join(PKG, "bin", "cilqr_replay")
    assert os.path.exists(exe), "bin/cilqr_replay not built (make -C %s)" % PKG
    N, P = 50, 200
    po = oracle.default_params(N)
    i = np.arange(float(P))
    path = np.stack([i, 0.5 * np.sin(0.05 * i)], axis=1)
    ego = np.array([0.0, 0.1, 3.0, 0.02])
    U = oracle.default_control_seq(N) if hasattr(oracle, "default_control_seq") else None
    if U is None:
        U = cilqr.default_control_seq(N)
    obst = {2: [(22.0, 0.9, 0.0, 0.1, 4.79, 2.16)], 3: [(22.0, 0.9, 0.0, 0.1, 4.79, 2.16), (40.0, -1.2, 0.0, -0.05, 4.2, 1.9)],
            4: [(40.0, -1.2, 0.0, -0.05, 4.2, 1.9)]}
    lines = ["# six ticks, closed loop on the oracle's own plan", "cilqr-replay 1", "horizon %d" % N, "path %d" % P]
    lines += ["%r %r" % (float(a), float(b)) for a, b in path]
    want = []
    for tick in range(6):
        obs = obst.get(tick, [])
        lines += ["tick", "ego " + " ".join(repr(float(v)) for v in ego), "obstacles %d" % len(obs)]
        lines += [" ".join(repr(float(v)) for v in o) for o in obs]
        coeffs, ref = oracle.local_plan(po, path, ego)
        M = len(obs)
        pose = np.array([[list(o[:4]) * 1 for _ in range(N)] for o in obs]).reshape(M, N, 4) if M else None
        dim = np.array([[list(o[4:]) for _ in range(N)] for o in obs]).reshape(M, N, 2) if M else None
        r = oracle.solve(po, N, ego, U, coeffs, ref[0, 0], ref[-1, 0], None if not M else pose.reshape(M, 4 * N),
                         None if not M else dim.reshape(M, 2 * N))
        want.append((ego.copy(), r))
        U = r["U"]
        ego = r["X"].reshape(N + 1, 4)[1].copy()
    log = tmp_path / "ticks.log"
    log.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, str(log)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 6
    for line, (e, r) in zip(out, want):
        w = line.split()
        assert w[0] == "experiment" and w[9] == "X" and w[10 + 4 * (N + 1)] == "U"
        assert np.array_equal(np.array(w[1:5], dtype=float), e)  # start_pos
        assert float(w[5]) > 0.0  # planning_time
        assert int(w[6]) == r["iters"] and int(w[7]) == r["status"]
        X = np.array(w[10:10 + 4 * (N + 1)], dtype=float)
        Uo = np.array(w[11 + 4 * (N + 1):], dtype=float)
        assert Uo.size == 2 * N
        assert np.max(np.abs(Uo - r["U"])) < TIGHT and np.max(np.abs(X - r["X"])) < 1e-7
    # malformed input is an error with a message, not a crash
    bad = tmp_path / "bad.log"
    bad.write_text("cilqr-replay 1\nhorizon 50\npath 2\n0 0\n")
    p = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert p.returncode == 1 and "replay log" in p.stderr


def test_config5_shard_full_size_properties(cilqr, oracle):
    """BASELINE config 5 at its per-GPU size (B = 8192, N = 80, M = 16; grouped family, G = 8), through properties that do
    not need 8192 oracle solves: (a) a 192-solve sample against the oracle; (b) solves are independent — a permuted batch
    gives the permuted outputs bit for bit, although every solve then sits in a different wavefront and lane group;
    (c) idempotence — the same call twice gives identical bits; (d) the held-row path (obstacles constant over the
    horizon are read from one table row) next to the streamed path: moving one obstacle by 1 cm in the last step of 64
    solves makes it time-varying there — those solves follow the oracle on the changed scene, all others keep their bits."""
    from cilqr_amd import scenes
    p = cilqr.default_params(80)
    sc = scenes.make_c5(8192, p)
    B, N, M = 8192, 80, 16
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        got = _gpu_batch(s, sc)
        again = _gpu_batch(s, sc)
        perm = np.random.default_rng(3).permutation(B)
        scp = dict(sc)
        for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"):
            scp[k] = np.ascontiguousarray(sc[k][perm])
        gotp = _gpu_batch(s, scp)
        # time-varying twin: step N-1 of obstacle 3 moved by 1 cm in the first 64 solves
        sct = dict(sc)
        pose = sc["obs_pose"].copy().reshape(B, M, N, 4)
        pose[:64, 3, N - 1, 0] += 0.01
        sct["obs_pose"] = pose.reshape(B, M, 4 * N)
        gott = _gpu_batch(s, sct)
    finally:
        s.close()
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(got[k], again[k]), k
        assert np.array_equal(got[k][perm], gotp[k]), k
        assert np.array_equal(got[k][64:], gott[k][64:]), k  # untouched solves: unchanged
    idx = np.concatenate([np.arange(64), np.arange(4000, 4064), np.arange(B - 64, B)])
    sub = {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), TIGHT, "c5 sample")
    subt = {k: (v[:64] if isinstance(v, np.ndarray) else v) for k, v in sct.items()}
    _compare({k: v[:64] for k, v in gott.items()}, _oracle_batch(oracle, N, subt), TIGHT, "c5 streamed obstacle")
    assert np.isfinite(got["U"]).all() and (got["iters"] >= 1).all()


# ---- sampled obstacles in compact form (BASELINE config 3) -----------------------------------------------------------------
def test_sampled_obstacles_match_oracle_and_materialised_call(cilqr, oracle):
    """cilqr_solve_batch_sampled: 8 moving obstacles x 32 pose samples given as nominal trajectories + offsets, against
    (a) the oracle on the materialised 256-obstacle scene (the reference's own Obstacle path with w_obstacle = 1/32,
    SURVEY §8c) and (b) the product's materialised call.  Sample headings come from angle addition and the semi-axes from
    refined reciprocals, so (b) is equality to TIGHT, not bit for bit.  Ragged shapes: n_obs = 3, S = 5, N = 30."""
    from cilqr_amd import scenes
    p = cilqr.default_params(50)
    sc = scenes.make_c3(96, p)
    s = cilqr.Solver(p, max_batch=96, max_horizon=50, max_obstacles=256, device=0)
    try:
        got = s.solve_batch_sampled(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                    sc["offsets"], sc["sample_weight"])
        mat = _gpu_batch(s, sc)
        # an offset of exactly zero in every sample must reproduce the nominal obstacle counted S times
        z = s.solve_batch_sampled(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                  np.zeros_like(sc["offsets"]), sc["sample_weight"])
        nominal = s.solve_batch(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                np.full((96, 8), 1.0))  # weight 32 x 1/32
        with pytest.raises(cilqr.CilqrError, match="-1"):  # one sample is not a sampled scene
            s.solve_batch_sampled(50, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                  sc["offsets"][:, :, :1], 1.0)
    finally:
        s.close()
    want = _oracle_batch(oracle, 50, sc)
    _compare(got, want, TIGHT, "sampled vs oracle")
    _compare(got, mat, TIGHT, "sampled vs materialised")
    _compare(z, nominal, TIGHT, "zero offsets vs nominal")
    # ragged: 3 obstacles x 5 samples, N = 30
    p30 = cilqr.default_params(30)
    rng = np.random.default_rng(8)
    base = scenes.make_static(16, 30, 3, p30, 77)
    off = rng.normal(0.0, 1.0, (16, 3, 5, 3)) * np.array([0.3, 0.3, 0.05])
    pose = base["obs_pose"].reshape(16, 3, 1, 30, 4).repeat(5, axis=2).copy()
    pose[..., 0] += off[..., 0][..., None]
    pose[..., 1] += off[..., 1][..., None]
    pose[..., 3] += off[..., 2][..., None]
    dim = base["obs_dim"].reshape(16, 3, 1, 30, 2).repeat(5, axis=2)
    s = cilqr.Solver(p30, max_batch=16, max_horizon=30, max_obstacles=15, device=0)
    try:
        got = s.solve_batch_sampled(30, base["x0"], base["U"], base["poly"], base["xplan_fl"], base["obs_pose"], base["obs_dim"],
                                    off, 0.2)
    finally:
        s.close()
    po = oracle.default_params(30)
    want = oracle.solve_batch(po, 30, 15, base["x0"], base["U"], base["poly"], base["xplan_fl"], pose.reshape(16, 15, 120),
                              dim.reshape(16, 15, 60), np.full((16, 15), 0.2), threads=min(16, oracle.max_threads()))
    _compare(got, want, TIGHT, "sampled ragged")


def test_long_horizons(cilqr, oracle):
    """Horizons whose per-solve arrays need more than the default 64 KiB of dynamic LDS (N = 256: ≈ 90 KiB) run after an
    explicit opt-in.  Beyond CILQR_MAX_HORIZON cilqr_create refuses; and a sampled-obstacle call whose offset records do not
    fit the CU's 160 KiB beside the solve is refused with CILQR_ERR_UNSUPPORTED — never a failed launch."""
    from cilqr_amd import scenes
    for N, M, B in ((150, 2, 6), (256, 3, 5)):
        p = cilqr.default_params(N)
        sc = scenes.make_static(B, N, M, p, 90 + N)
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
        try:
            got = _gpu_batch(s, sc)
        finally:
            s.close()
        _compare(got, _oracle_batch(oracle, N, sc), 1e-8, "N%d" % N)
    with pytest.raises(cilqr.CilqrError, match="-1"):
        cilqr.Solver(cilqr.default_params(600), max_batch=2, max_horizon=600, max_obstacles=0, device=0)
    # sampled obstacles: 8 x 256 offset records (64 KB) beside a 384-step solve (132 KB) do not fit one CU's LDS
    p = cilqr.default_params(384)
    sc = scenes.make_static(2, 384, 8, p, 3)
    s = cilqr.Solver(p, max_batch=2, max_horizon=384, max_obstacles=2048, device=0)
    try:
        with pytest.raises(cilqr.CilqrError, match="-4"):
            s.solve_batch_sampled(384, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"],
                                  np.zeros((2, 8, 256, 3)), 1.0 / 256)
    finally:
        s.close()


@pytest.mark.parametrize("G", [0, 8])
def test_closest_point_windows_on_steep_and_distant_paths(cilqr, oracle, G, monkeypatch):
    """The closest-sample search prunes with two windows (x side, y side; cilqr_device.hpp::closest_sample); both must return
    exactly the reference's argmin over all 200 samples.  Stress them away from the benchmark's gentle paths: slopes up
    to ±2 (the y-side bound must switch itself off), flat paths (adjacent samples equal in y), egos up to 6 m beside the
    path and beyond either end of it — whole solves against the oracle, both kernel families."""
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 50, 2, 192
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 4242)
    rng = np.random.default_rng(4243)
    poly = np.zeros((B, 6))
    slope = rng.uniform(-2.0, 2.0, B)
    slope[:32] = 0.0                                   # flat: D = 0
    curv = rng.uniform(-0.02, 0.02, B)
    curv[:48] = 0.0
    xf = sc["xplan_fl"][:, 0]
    # y(x) = y0 + slope (x - xf) + curv (x - xf)^2, expanded in powers of x
    y0 = rng.uniform(-3, 3, B)
    poly[:, 0] = y0 - slope * xf + curv * xf * xf
    poly[:, 1] = slope - 2 * curv * xf
    poly[:, 2] = curv
    x0 = sc["x0"].copy()
    along = rng.uniform(-8.0, 28.0, B)                 # before the first sample … beyond the last (the plan spans ≈ 19 m)
    x0[:, 0] = xf + along
    x0[:, 1] = y0 + slope * along + curv * along * along + rng.uniform(-6.0, 6.0, B)
    x0[:, 3] = np.arctan(slope) + rng.uniform(-0.3, 0.3, B)
    sc2 = dict(sc, poly=poly, x0=x0)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        got = _gpu_batch(s, sc2)
    finally:
        s.close()
    want = _oracle_batch(oracle, N, sc2)
    ok = np.isfinite(want["U"]).all(axis=1)
    assert ok.sum() >= B * 0.9
    _compare({k: v[ok] for k, v in got.items()}, {k: v[ok] for k, v in want.items()}, 1e-8, "closest-point windows G=%d" % G)
    # scenes the oracle leaves non-finite (a far ego on a steep path overflows the barrier exponentials): the kernels must
    # report them the same way — same exit reason, same iteration count, no finite-looking trajectory
    bad = ~ok
    assert np.array_equal(got["status"][bad], want["status"][bad])
    assert np.array_equal(got["iters"][bad], want["iters"][bad])
    assert (~np.isfinite(got["U"][bad]).all(axis=1) | (got["status"][bad] == cilqr.EXIT_NUMERIC)).all()


def test_closest_sample_search_equals_full_scan(cilqr, solver):
    """`cilqr_debug_closest_sample`: the kernels' closest-sample search (two pruning windows; Newton on the continuous distance where
    the window is wide and the distance provably convex over it) against a plain scan over all 200 samples in the same kernel, on
    400 000 random queries: gentle paths as the benchmarks have them, steep and strongly curved ones, cubic and quintic terms, points
    on the path, far beside it (both sides of the centre of curvature), before and beyond its ends, reversed sample order, exact
    ties (a straight path with the point midway between two samples).  Every index equal; and the Newton search must really be
    the one that decides a good share of the wide-window queries (else this test tests the scan against itself)."""
    rng = np.random.default_rng(9100)
    n = 400_000
    q = np.zeros((n, 10))
    xf = rng.uniform(-5.0, 5.0, n)
    length = rng.uniform(8.0, 30.0, n) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
    kind = rng.integers(0, 4, n)
    slope = np.where(kind == 0, rng.uniform(-0.3, 0.3, n), rng.uniform(-2.0, 2.0, n))
    curv = np.where(kind == 0, rng.uniform(-0.01, 0.01, n), np.where(kind == 1, 0.0, rng.uniform(-0.2, 0.2, n)))
    c3 = np.where(kind == 3, rng.uniform(-0.004, 0.004, n), 0.0)
    c5 = np.where(kind == 3, rng.uniform(-2e-6, 2e-6, n), 0.0)
    y0 = rng.uniform(-3.0, 3.0, n)
    # y(u) = y0 + slope u + curv u² + c3 u³ + c5 u⁵, u = x - xf, expanded in powers of x (binomial sums, vectorised)
    cu = np.stack([y0, slope, curv, c3, np.zeros(n), c5], axis=1)
    from math import comb
    for j in range(6):
        for i in range(j + 1):
            q[:, i] += cu[:, j] * comb(j, i) * (-xf) ** (j - i)
    q[:, 6] = xf
    q[:, 7] = xf + length
    along = rng.uniform(-0.3, 1.3, n) * length
    yp = y0 + slope * along + curv * along ** 2 + c3 * along ** 3 + c5 * along ** 5
    lateral = np.where(rng.random(n) < 0.2, 0.0, rng.uniform(-10.0, 10.0, n))
    q[:, 8] = xf + along
    q[:, 9] = yp + lateral
    # exact ties: a horizontal straight path, the point above the midpoint of two samples (dxs = length / 200, exactly representable)
    t = np.arange(0, 2000)
    q[t, :6] = 0.0
    q[t, 0] = 1.0
    q[t, 6] = 0.0
    q[t, 7] = 25.0
    q[t, 8] = 0.125 * (t % 190) + 0.0625
    q[t, 9] = 1.0 + rng.uniform(0.0, 7.0, len(t))
    out = solver.debug_closest_sample(q)
    bad = np.nonzero(out[:, 0] != out[:, 1])[0]
    assert bad.size == 0, "search != full scan at %d queries, first: %s -> %s" % (bad.size, q[bad[:1]], out[bad[:1]])
    share = out[:, 2].mean()
    print("closest-sample search: Newton decided %.1f %% of %d queries" % (100 * share, n))
    assert share > 0.15


@pytest.mark.parametrize("G", [0, 8])
def test_closest_point_newton_on_curved_paths(cilqr, oracle, G, monkeypatch):
    """Wide search windows are resolved by Newton on the continuous distance where it is provably convex over the window
    (cilqr_device.hpp::closest_newton), by the scan otherwise — the reference's argmin over all 200 samples either way.  Paths with
    real curvature (second-order coefficient up to ±0.2: radius 2.5 m, egos on both sides of the centre of curvature, where convexity
    fails and the search must fall back), cubic and quintic terms (the bound of the second derivative is taken over all samples),
    egos up to 8 m beside the path and beyond its ends — whole solves against the oracle; the shared-phase-L kernel (G = 0) and the
    grouped family."""
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 40, 2, 256
    p = cilqr.default_params(N)
    sc = scenes.make_static(B, N, M, p, 4342)
    rng = np.random.default_rng(4343)
    xf = sc["xplan_fl"][:, 0]
    slope = rng.uniform(-1.0, 1.0, B)
    curv = rng.uniform(-0.2, 0.2, B)
    curv[:64] = rng.uniform(-0.03, 0.03, 64)
    c3 = rng.uniform(-0.004, 0.004, B)
    c5 = rng.uniform(-2e-6, 2e-6, B)
    c3[64:128] = 0.0
    c5[64:160] = 0.0
    y0 = rng.uniform(-3, 3, B)
    # y(u) = y0 + slope u + curv u² + c3 u³ + c5 u⁵ with u = x - xf, expanded in powers of x
    poly = np.zeros((B, 6))
    for b in range(B):
        q = np.polynomial.polynomial.Polynomial([y0[b], slope[b], curv[b], c3[b], 0.0, c5[b]])
        shifted = q(np.polynomial.polynomial.Polynomial([-xf[b], 1.0]))
        co = shifted.coef
        poly[b, :len(co)] = co
    x0 = sc["x0"].copy()
    along = rng.uniform(-6.0, 26.0, B)
    yp = y0 + slope * along + curv * along ** 2 + c3 * along ** 3 + c5 * along ** 5
    x0[:, 0] = xf + along
    x0[:, 1] = yp + rng.uniform(-8.0, 8.0, B)
    x0[:, 3] = np.arctan(slope + 2 * curv * along) + rng.uniform(-0.3, 0.3, B)
    sc2 = dict(sc, poly=poly, x0=x0)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        got = _gpu_batch(s, sc2)
        if not G:
            assert s.solve_wavefronts(B, N, M) == 3
    finally:
        s.close()
    want = _oracle_batch(oracle, N, sc2)
    ok = np.isfinite(want["U"]).all(axis=1)
    assert ok.sum() >= B * 0.75
    _compare({k: v[ok] for k, v in got.items()}, {k: v[ok] for k, v in want.items()}, 1e-8, "closest point by Newton G=%d" % G)
    bad = ~ok
    assert np.array_equal(got["status"][bad], want["status"][bad])
    assert np.array_equal(got["iters"][bad], want["iters"][bad])


@pytest.mark.parametrize("G", [1, 8, 32])
def test_early_exit_equals_reference_loop_grouped_family(cilqr, monkeypatch, G):
    """The same equivalence for the G-lanes-per-solve family: stopping at the first rejection and replaying the λ / counter
    arithmetic gives bit for bit what executing the rejected iterations' passes gives.  (With the flag, a solve that has
    rejected stops swapping its trajectory buffers while its neighbours in the wavefront go on: the forward pass must then
    read each group's own buffer — this test found the staged copy using one source for the whole wavefront.)"""
    from cilqr_amd import scenes
    monkeypatch.setenv("CILQR_FORCE_G", str(G))
    p = cilqr.default_params(50)
    sc = scenes.make_static(200, 50, 4, p, 515)
    s = cilqr.Solver(p, max_batch=200, max_horizon=50, max_obstacles=4, device=0)
    try:
        a, b = _gpu_batch(s, sc), _gpu_batch(s, sc, flags=cilqr.FLAG_FAITHFUL_ITERS)
    finally:
        s.close()
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(a[k], b[k]), k


def test_device_pointer_entry_points_on_a_side_stream(cilqr):
    """The *_device entry points (device pointers, caller's stream, asynchronous) give bit for bit what the host-buffer entry
    points give: plain solve, sampled solve (also under FAITHFUL_ITERS) and the batched LocalPlanner, enqueued back to
    back on a non-default stream with a single synchronisation at the end."""
    import torch
    from cilqr_amd import scenes
    dev = torch.device("cuda", 0)
    N, B = 50, 96
    p = cilqr.default_params(N)
    sc = scenes.make_c3(B, p, n_dyn=4, n_samples=8)
    M = sc["M"]
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        host_plain = s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"])
        host_samp = s.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                          sc["offsets"], sc["sample_weight"])
        # (the reference loop runs on the one-wavefront kernel, the early exit on two wavefronts per solve that add a step's
        # obstacle terms in another order: equal to rounding, test_split_kernel_…; bit for bit each against its own host call)
        host_samp_f = s.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"],
                                            sc["offsets"], sc["sample_weight"], flags=cilqr.FLAG_FAITHFUL_ITERS)
        assert np.array_equal(host_samp["iters"], host_samp_f["iters"]) and np.max(np.abs(host_samp["U"] - host_samp_f["U"])) < 1e-11
        i = np.arange(200.0)
        path = np.stack([i, 0.5 * np.sin(0.05 * i)], axis=1)
        egos = np.stack([np.linspace(0, 150, B), 0.5 * np.sin(0.05 * np.linspace(0, 150, B)) + 0.2, np.full(B, 3.0), np.zeros(B)], axis=1)
        host_plan = s.local_plan_batch(path, egos)

        t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)  # noqa: E731
        d = {k: t(sc[k]) for k in ("x0", "poly", "xplan_fl", "obs_pose", "obs_dim", "obs_weight", "nom_pose", "nom_dim", "offsets")}
        U1, U2, U3 = t(sc["U"]), t(sc["U"]), t(sc["U"])
        outs = [dict(X=torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev), J=torch.zeros(B, dtype=torch.float64, device=dev),
                     it=torch.zeros(B, dtype=torch.int32, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(3)]
        d_path, d_ego = t(path), t(egos)
        d_poly = torch.zeros(B, 6, dtype=torch.float64, device=dev)
        d_fl = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            st_ = side.cuda_stream
            o = outs[0]
            s.solve_batch_device(st_, B, N, M, d["x0"].data_ptr(), U1.data_ptr(), d["poly"].data_ptr(), d["xplan_fl"].data_ptr(),
                                 d["obs_pose"].data_ptr(), d["obs_dim"].data_ptr(), d["obs_weight"].data_ptr(), o["X"].data_ptr(),
                                 o["J"].data_ptr(), o["it"].data_ptr(), o["st"].data_ptr())
            for o, U, fl in ((outs[1], U2, 0), (outs[2], U3, cilqr.FLAG_FAITHFUL_ITERS)):
                s.solve_batch_sampled_device(st_, B, N, 4, 8, d["x0"].data_ptr(), U.data_ptr(), d["poly"].data_ptr(),
                                             d["xplan_fl"].data_ptr(), d["nom_pose"].data_ptr(), d["nom_dim"].data_ptr(),
                                             d["offsets"].data_ptr(), sc["sample_weight"], o["X"].data_ptr(), o["J"].data_ptr(),
                                             o["it"].data_ptr(), o["st"].data_ptr(), flags=fl)
            s.local_plan_batch_device(st_, B, 200, d_path.data_ptr(), 0, d_ego.data_ptr(), d_poly.data_ptr(), d_fl.data_ptr())
        side.synchronize()
    finally:
        s.close()
    for o, U, want in ((outs[0], U1, host_plain), (outs[1], U2, host_samp), (outs[2], U3, host_samp_f)):
        assert np.array_equal(U.cpu().numpy(), want["U"])
        assert np.array_equal(o["X"].cpu().numpy(), want["X"]) and np.array_equal(o["J"].cpu().numpy(), want["J"])
        assert np.array_equal(o["it"].cpu().numpy(), want["iters"]) and np.array_equal(o["st"].cpu().numpy(), want["status"])
    assert np.array_equal(d_poly.cpu().numpy(), host_plan["poly"]) and np.array_equal(d_fl.cpu().numpy(), host_plan["xplan_fl"])


def test_occupancy_conversion_odd_ranges_large_grid(cilqr, oracle, solver):
    """Ranges for which the step-table path must stand aside (reversed, empty, denormal, infinite or NaN bounds) and ordinary
    ones, on a grid large enough (2^20 cells) for that path to be considered: always the oracle's bytes."""
    rng = np.random.default_rng(21)
    n = 1 << 20
    vals = rng.uniform(-50, 150, n).astype(np.float32)
    vals[rng.integers(0, n, 5000)] = np.nan
    vals[rng.integers(0, n, 3000)] = np.float32(np.inf)
    vals[rng.integers(0, n, 3000)] = np.float32(-np.inf)
    vals[:101] = np.arange(101, dtype=np.float32)
    for lo, hi in ((0.0, 100.0), (100.0, 0.0), (7.0, 7.0), (0.0, 1e-40), (0.0, np.inf), (-np.inf, 0.0), (np.nan, 1.0), (0.0, np.nan),
                   (-3.0e38, 3.0e38), (1e-3, 2e-3), (-1.0, 100.0)):
        got = solver.layer_to_occupancy(vals, lo, hi)
        want = oracle.layer_to_occupancy(vals, lo, hi)
        assert np.array_equal(got, want), (lo, hi, int((got != want).sum()))


def test_local_plan_batch_other_slice_length(cilqr, oracle):
    """num_of_local_wpts ≠ 20 takes the plain-loop instantiation of the fit kernel (and changes the solver's path-sample
    count to 10 × that): fit and a plan + solve against the oracle with num_of_local_wpts = 12."""
    from cilqr_amd import scenes
    N, B = 40, 64
    p = cilqr.default_params(N)
    p.num_of_local_wpts = 12
    po = oracle.default_params(N)
    po.num_of_local_wpts = 12
    i = np.arange(120.0)
    path = np.stack([i, 0.8 * np.sin(0.04 * i)], axis=1)
    rng = np.random.default_rng(31)
    along = rng.uniform(0, 118, B)
    egos = np.stack([along, 0.8 * np.sin(0.04 * along) + rng.uniform(-0.5, 0.5, B), rng.uniform(1, 6, B), rng.uniform(-0.1, 0.1, B)], axis=1)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=0, device=0)
    try:
        plan = s.local_plan_batch(path, egos)
        got = s.solve_batch(N, egos, np.tile(cilqr.default_control_seq(N), (B, 1)), plan["poly"], plan["xplan_fl"])
    finally:
        s.close()
    poly, fl, n, refs = _oracle_plans(oracle, po, path, egos)
    assert np.array_equal(plan["n"], n) and n.max() == 12
    assert np.array_equal(plan["poly"], poly) and np.array_equal(plan["xplan_fl"], fl)  # integer abscissae: exact powers
    want = oracle.solve_batch(po, N, 0, egos, np.tile(cilqr.default_control_seq(N), (B, 1)), poly, fl, None, None, None,
                              threads=min(16, oracle.max_threads()))
    _compare(got, want, TIGHT, "12 waypoints")


@pytest.mark.parametrize("G", [0, 8])
def test_non_default_parameters(cilqr, oracle, G, monkeypatch):
    """Every field of cilqr_params reaches the kernels: short iteration caps, a loose tolerance (exit by tolerance, status 0),
    other weights, barrier constants, limits, margins, timestep and regularisation schedule — both kernel families."""
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 40, 3, 96
    variants = [
        dict(max_iterations=6),
        dict(tolerance=0.5),
        dict(max_iterations=33, lamb_factor=4.0, lamb_max=300.0),
        dict(w_acc=2.0, w_yawrate=1.5, w_pos=1.1, w_vel=0.7, w_obstacle=0.6, desired_speed=7.5),
        dict(q1_acc=0.7, q2_acc=1.3, q1_yawrate=1.2, q2_yawrate=0.8, q1_front=2.0, q2_front=2.2, q1_rear=3.0, q2_rear=1.9),
        dict(acc_max=1.2, acc_min=-3.0, steer_angle_max=0.5, steer_angle_min=-0.4, wheelbase=2.5, speed_max=12.0),
        dict(t_safe=0.4, s_safe_a=0.3, s_safe_b=0.2, ego_rad=1.1, ego_front=1.8, ego_rear=2.1, timestep=0.07),
    ]
    saw_tolerance_exit = False
    for k, v in enumerate(variants):
        p, po = cilqr.default_params(N), oracle.default_params(N)
        for name, val in v.items():
            assert hasattr(p, name) and hasattr(po, name), name
            setattr(p, name, val)
            setattr(po, name, val)
        sc = scenes.make_static(B, N, M, p, 700 + k)
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
        try:
            got = _gpu_batch(s, sc)
        finally:
            s.close()
        want = oracle.solve_batch(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None,
                                  threads=min(16, oracle.max_threads()))
        _compare(got, want, 1e-8, "variant %d G=%d" % (k, G))
        saw_tolerance_exit |= bool((want["status"] == 0).any())
        if "max_iterations" in v:
            assert want["iters"].max() <= v["max_iterations"]
    assert saw_tolerance_exit


def test_two_handles_overlapping_on_two_streams(cilqr):
    """Solves that overlap in time need separate handles (include/cilqr.h): two handles, two streams, launches interleaved
    without synchronising in between — each must give what it gives alone."""
    import torch
    from cilqr_amd import scenes
    dev = torch.device("cuda", 0)
    N, M, B = 50, 4, 512
    p = cilqr.default_params(N)
    scs = [scenes.make_static(B, N, M, p, 900 + k) for k in range(2)]
    solvers = [cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0) for _ in range(2)]
    try:
        alone = [s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"]) for s, sc in zip(solvers, scs)]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        dv = [{k: t(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")} for sc in scs]
        X = [torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev) for _ in range(2)]
        J = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(2)]
        it = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
        st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
        U0 = [d["U"].clone() for d in dv]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(3):  # the same solve three times per handle, streams interleaved
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    dv[k]["U"].copy_(U0[k])
                    solvers[k].solve_batch_device(streams[k].cuda_stream, B, N, M, dv[k]["x0"].data_ptr(), dv[k]["U"].data_ptr(),
                                                  dv[k]["poly"].data_ptr(), dv[k]["xplan_fl"].data_ptr(), dv[k]["obs_pose"].data_ptr(),
                                                  dv[k]["obs_dim"].data_ptr(), 0, X[k].data_ptr(), J[k].data_ptr(), it[k].data_ptr(),
                                                  st[k].data_ptr())
        for s_ in streams:
            s_.synchronize()
    finally:
        for s in solvers:
            s.close()
    for k in range(2):
        assert np.array_equal(dv[k]["U"].cpu().numpy(), alone[k]["U"]) and np.array_equal(X[k].cpu().numpy(), alone[k]["X"])
        assert np.array_equal(it[k].cpu().numpy(), alone[k]["iters"]) and np.array_equal(st[k].cpu().numpy(), alone[k]["status"])


def test_local_plan_batch_tiny_paths(cilqr, oracle, solver):
    """Paths of 1, 2, 3 and 7 waypoints (fewer rows than the six unknowns, down to a single row), one candidate or several,
    shared or per-candidate paths: slice, coefficients and fitted reference against the oracle (integer abscissae: exact)."""
    p = oracle.default_params(50)
    rng = np.random.default_rng(41)
    for P in (1, 2, 3, 7):
        for B in (1, 5):
            x = np.arange(P, dtype=float) + 3.0
            path = np.stack([x, 0.25 * x - 1.0 + 0.1 * np.sin(x)], axis=1)
            egos = np.stack([rng.uniform(2, 3 + P, B), rng.uniform(-1, 1, B), np.full(B, 2.0), np.zeros(B)], axis=1)
            got = solver.local_plan_batch(path, egos)
            poly, fl, n, refs = _oracle_plans(oracle, p, path, egos)
            assert np.array_equal(got["n"], n) and np.array_equal(got["xplan_fl"], fl), (P, B)
            assert np.array_equal(got["poly"], poly), (P, B)
            for b in range(B):
                assert np.array_equal(got["ref_traj"][b, :n[b]], refs[b]), (P, B, b)
            per = np.broadcast_to(path, (B, P, 2)).copy()
            got2 = solver.local_plan_batch(per, egos)
            assert np.array_equal(got2["poly"], poly) and np.array_equal(got2["n"], n), (P, B)


def test_costmap_kernels_on_tiny_and_odd_maps(cilqr, oracle, solver):
    """Warp, blur and conversions on maps of 1x1, 3x2, 65x17 cells (tiles, wavefronts and the conversions' 4-cell groups all
    partially filled) and a destination mostly outside its source: every output and every out-of-range count as the oracle."""
    rng = np.random.default_rng(51)
    for (slx, sly, sres), (dlx, dly, dres, dpx, dpy), pose in [
            ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0, 0.0, 0.0), (0.1, -0.1, 0.3)),
            ((3.0, 2.0, 1.0), (3.0, 2.0, 1.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
            ((13.0, 3.4, 0.2), (6.5, 1.7, 0.1, 0.3, -0.2), (0.4, 0.1, 2.0)),
            ((4.0, 4.0, 0.5), (20.0, 12.0, 0.5, 1.0, 1.0), (0.5, 0.5, 0.7))]:   # destination far larger than the source
        sg, dg = cilqr.map_geom(slx, sly, sres, 0.5, -0.5), cilqr.map_geom(dlx, dly, dres, dpx, dpy)
        osg, odg = oracle.map_geom(slx, sly, sres, 0.5, -0.5), oracle.map_geom(dlx, dly, dres, dpx, dpy)
        src = np.asfortranarray(rng.integers(0, 101, (sg.rows, sg.cols)).astype(np.float32))
        if src.size > 4:
            src[rng.random(src.shape) < 0.1] = np.nan
        got, n_oob = solver.warp_costmap(src, sg, dg, *pose)
        want, w_oob = oracle.warp(src, osg, odg, *pose)
        assert n_oob == w_oob and np.array_equal(got, want, equal_nan=True), (sg.rows, sg.cols, dg.rows, dg.cols)
        blur_in = np.where(np.isnan(want), np.float32(0), want).astype(np.float32)
        b_got = solver.blur_costmap(blur_in, dg, pose[2], 0.16, 0.16, 0.017)
        b_got = b_got[0] if isinstance(b_got, tuple) else b_got
        b_want, _, _ = oracle.blur(blur_in, odg, np.sin(pose[2]), np.cos(pose[2]), 0.16, 0.16, 0.017)
        d = _ulp32_diff(np.asarray(b_got, dtype=np.float32).reshape(-1, order="F"), b_want.reshape(-1, order="F"))
        assert d.max() <= 1, (dg.rows, dg.cols, int(d.max()))
        occ = solver.layer_to_occupancy(b_want.reshape(-1, order="F"), 0.0, 100.0)
        assert np.array_equal(occ, oracle.layer_to_occupancy(b_want.reshape(-1, order="F"), 0.0, 100.0))
        assert np.array_equal(solver.occupancy_to_layer(occ), oracle.occupancy_to_layer(occ), equal_nan=True)


def test_sampled_obstacles_odd_counts_long_horizon(cilqr, oracle):
    """Sampled obstacles with 33 samples (not a power of two), a single obstacle and 7 obstacles, horizon 80, one of the
    obstacles far away for the whole horizon (the whole-obstacle skip) — against the oracle on the materialised scene."""
    from cilqr_amd import scenes
    N, B = 80, 24
    p = cilqr.default_params(N)
    po = oracle.default_params(N)
    rng = np.random.default_rng(61)
    for n_obs, S in ((1, 33), (7, 3)):
        base = scenes.make_static(B, N, n_obs, p, 800 + n_obs)
        nom_pose = base["obs_pose"].reshape(B, n_obs, N, 4).copy()
        nom_pose[:, -1, :, 0] += 500.0  # the last obstacle: half a kilometre away
        # let the obstacles move: x advances with the step index
        nom_pose[:, :, :, 0] += 0.2 * np.arange(N)[None, None, :]
        off = rng.normal(0.0, 1.0, (B, n_obs, S, 3)) * np.array([0.2, 0.2, 0.03])
        pose = np.repeat(nom_pose[:, :, None, :, :], S, axis=2)
        pose[..., 0] += off[..., 0][..., None]
        pose[..., 1] += off[..., 1][..., None]
        pose[..., 3] += off[..., 2][..., None]
        dim = np.repeat(base["obs_dim"].reshape(B, n_obs, 1, N, 2), S, axis=2)
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=n_obs * S, device=0)
        try:
            got = s.solve_batch_sampled(N, base["x0"], base["U"], base["poly"], base["xplan_fl"], nom_pose.reshape(B, n_obs, 4 * N),
                                        base["obs_dim"], off, 1.0 / S)
        finally:
            s.close()
        want = oracle.solve_batch(po, N, n_obs * S, base["x0"], base["U"], base["poly"], base["xplan_fl"],
                                  pose.reshape(B, n_obs * S, 4 * N), dim.reshape(B, n_obs * S, 2 * N), np.full((B, n_obs * S), 1.0 / S),
                                  threads=min(16, oracle.max_threads()))
        _compare(got, want, TIGHT, "sampled n_obs=%d S=%d" % (n_obs, S))


def test_config3_full_batch_properties(cilqr, oracle):
    """BASELINE config 3 at its full size (B = 4096, N = 50, 8 moving obstacles x 32 samples, compact entry point) through
    size-independent properties: a 128-solve sample against the oracle on the materialised 256-obstacle scene; permutation
    equivariance (every solve then runs in another workgroup); idempotence."""
    from cilqr_amd import scenes
    N, B = 50, 4096
    p = cilqr.default_params(N)
    sc = scenes.make_c3(B, p)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=sc["M"], device=0)

    def run(c):
        return s.solve_batch_sampled(N, c["x0"], c["U"], c["poly"], c["xplan_fl"], c["nom_pose"], c["nom_dim"], c["offsets"],
                                     c["sample_weight"])
    try:
        got = run(sc)
        again = run(sc)
        perm = np.random.default_rng(5).permutation(B)
        scp = dict(sc)
        for k in ("x0", "U", "poly", "xplan_fl", "nom_pose", "nom_dim", "offsets"):
            scp[k] = np.ascontiguousarray(sc[k][perm])
        gotp = run(scp)
    finally:
        s.close()
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(got[k], again[k]), k
        assert np.array_equal(got[k][perm], gotp[k]), k
    idx = np.concatenate([np.arange(48), np.arange(2000, 2040), np.arange(B - 40, B)])
    sub = {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    _compare({k: v[idx] for k, v in got.items()}, _oracle_batch(oracle, N, sub), 1e-9, "c3 full batch sample")
    assert np.isfinite(got["U"]).all()


def test_pass_count_buffer(cilqr, oracle):
    """cilqr_set_pass_count_buffer: executed backward+forward passes per solve.  With the early exit a solve that stops on
    a rejection after r accepted iterations executed r passes; one that stops on the tolerance executed `iters` passes; under
    CILQR_FLAG_FAITHFUL_ITERS every reference iteration is a pass.  Both kernel families."""
    import torch
    from cilqr_amd import scenes
    N, M = 50, 4
    p = cilqr.default_params(N)
    for B in (96, 8192):  # wavefront family; grouped family
        sc = scenes.make_static(B, N, M, p, 777)
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
        try:
            assert (s.solve_family(B, N, M) == 64) == (B == 96)
            dev = torch.device("cuda", 0)
            t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")}
            X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev)
            J = torch.zeros(B, dtype=torch.float64, device=dev)
            it = torch.zeros(B, dtype=torch.int32, device=dev)
            st = torch.zeros(B, dtype=torch.int32, device=dev)
            ps = torch.full((B,), -1, dtype=torch.int32, device=dev)
            s.set_pass_count_buffer(ps.data_ptr())
            stream = torch.cuda.current_stream().cuda_stream
            out = {}
            for flags in (0, cilqr.FLAG_FAITHFUL_ITERS):
                U = t["U"].clone()
                s.solve_batch_device(stream, B, N, M, t["x0"].data_ptr(), U.data_ptr(), t["poly"].data_ptr(), t["xplan_fl"].data_ptr(),
                                     t["obs_pose"].data_ptr(), t["obs_dim"].data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(),
                                     st.data_ptr(), flags)
                torch.cuda.synchronize()
                out[flags] = (ps.cpu().numpy().copy(), it.cpu().numpy().copy(), st.cpu().numpy().copy())
            s.set_pass_count_buffer(0)
        finally:
            s.close()
        ps0, it0, st0 = out[0]
        psf, itf, stf = out[cilqr.FLAG_FAITHFUL_ITERS]
        assert np.array_equal(it0, itf) and np.array_equal(st0, stf)
        tol = st0 == cilqr.EXIT_TOLERANCE
        assert np.array_equal(ps0[tol], it0[tol])
        # a rejection after r accepted iterations: the reference loop then only multiplies lamb (1/10^r by repeated division) by
        # 10 per iteration until it passes lamb_max or the iteration cap — replayed here in the same double arithmetic
        def replay(r):
            lamb = 1.0
            for _ in range(r):
                lamb = lamb / p.lamb_factor
            k = r + 1
            while True:
                lamb = lamb * p.lamb_factor
                if lamb > p.lamb_max:
                    return k, cilqr.EXIT_LAMBDA_MAX
                if k >= p.max_iterations:
                    return k, cilqr.EXIT_MAX_ITER
                k += 1
        rej = ~tol & (st0 != cilqr.EXIT_NUMERIC)
        want = np.array([replay(int(r)) for r in ps0[rej]]).reshape(-1, 2)
        full = ps0[rej] == p.max_iterations  # all max_iterations iterations accepted: no rejection at all
        assert np.array_equal(want[~full, 0], it0[rej][~full]) and np.array_equal(want[~full, 1], st0[rej][~full])
        assert (ps0 >= 0).all() and (ps0 <= it0).all()
        # faithful loop: every iteration runs both passes, except the last when it ends on lamb > lamb_max after its passes
        assert (psf >= ps0).all() and (psf <= itf).all() and (psf[~tol] >= itf[~tol] - 1).all()


def test_run_step_and_run_candidates_agree(cilqr, oracle, tmp_path):
    """iLQR::run_step uses the host pre-step (libm pow in the Vandermonde matrix), run_candidates the device fit (correctly
    rounded powers): the two local plans differ in the last bits about once per thousand fits.  One candidate through either
    route must give the same trajectory to well within the solver tolerance."""
    import os
    import subprocess
    from conftest import PKG, ROOT
    exe = str(tmp_path / "step_vs_candidates")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "step_vs_candidates.cpp"), "-L" + os.path.join(PKG, "lib"), "-lcilqr_hip",
                    "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    worst = float(r.stdout.strip().split()[-1])
    assert worst <= 1e-9, r.stdout


# ---- the cross-GPU exchange step behind the C-ABI (csrc/cilqr_comm.cpp) ------------------------------------------------------
def test_cross_rank_select_rule_on_device(cilqr, solver):
    """select_kernel — the pick over the gathered per-rank records — on the cases the world-size-2 gloo test runs through
    the torch path (tests/test_multi_rank.py): rank without a finite cost, NaN, ties resolved by the lowest GLOBAL index."""
    from test_multi_rank import CASES, EXPECT
    for (per_rank, shard), (ej, ei) in zip(CASES, EXPECT):
        triples = [[p[0], p[1], r * shard] for r, p in enumerate(per_rank)]
        j, i = solver.debug_select(triples)
        assert i == ei and (j == ej or (j != j and ej != ej)), (triples, j, i)
    rng = np.random.default_rng(17)  # 70 ranks (more than one wavefront's worth of records), duplicated minima
    J = rng.integers(0, 9, 70).astype(float)
    idx = rng.integers(0, 50, 70).astype(float)
    off = np.arange(70) * 50.0
    j, i = solver.debug_select(np.stack([J, idx, off], 1))
    glob = idx + off
    k = np.lexsort((glob, J))[0]
    assert (j, i) == (J[k], int(glob[k]))


def test_argmin_global_device_single_rank_and_rccl_world1(cilqr, solver):
    """cilqr_argmin_global_device: without a communicator it is the local pick plus the offset; with a one-rank RCCL
    communicator (unique id → ncclCommInitRank → ncclAllGather, all through the C-ABI) the result is the same."""
    import torch
    dev = torch.device("cuda", 0)
    J = torch.tensor([5.0, float("nan"), 2.5, 7.0, 2.5, float("inf")], dtype=torch.float64, device=dev)
    pair = torch.zeros(2, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    solver.argmin_global_device(stream, 6, J.data_ptr(), 1000, pair.data_ptr())
    assert pair.tolist() == [2.5, 1002.0]
    solver.argmin_global_device(stream, 0, 0, 1000, pair.data_ptr())  # a rank without scenes
    assert pair.tolist() == [float("inf"), -1.0]
    s = cilqr.Solver(cilqr.default_params(), max_batch=8, max_horizon=8, max_obstacles=0, device=0)
    try:
        assert s.comm_size() == 1
        s.comm_init_rank(1, 0, cilqr.comm_unique_id())
        assert s.comm_size() == 1
        s.argmin_global_device(stream, 6, J.data_ptr(), 64, pair.data_ptr())
        assert pair.tolist() == [2.5, 66.0]
        with pytest.raises(cilqr.CilqrError):
            s.comm_init_rank(1, 0, cilqr.comm_unique_id())  # already has a communicator
    finally:
        s.close()


def _build_multi_device(tmp_path):
    import os
    import subprocess
    from conftest import PKG, ROOT
    exe = str(tmp_path / "multi_device")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "multi_device.cpp"), "-L" + os.path.join(PKG, "lib"), "-lcilqr_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def _run_json(cmd, env=None):
    import json
    import subprocess
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]  # RCCL may print a version banner before it
    assert r.returncode == 0 and lines, r.stdout + r.stderr
    return json.loads(lines[-1])


def test_cpp_host_shards_over_all_devices(cilqr, tmp_path):
    """tests/cpp/multi_device.cpp: a C++ host with no Python in it runs one batch over n = 1 .. cilqr_device_count() devices
    through cilqr_create_multi / cilqr_multi_solve_batch (RCCL exchange inside) — bit-equal to the single-handle batch and pick.
    Small batches take the packed staging path, B = 1500 the array-by-array copies, from pageable and from pinned memory."""
    exe = _build_multi_device(tmp_path)
    n = cilqr.lib().cilqr_device_count()
    assert n >= 1
    for args in (["--B", "37"], ["--B", "5"], ["--B", "1500"], ["--B", "1500", "--pinned"]):
        out = _run_json([exe] + args)
        assert out["bit_equal"] and out["devices"] == n and out["cases"] == 3 * n and out["rccl_cases"] == n, out


def test_cpp_host_shard_arithmetic_with_several_shards_on_one_device(cilqr, tmp_path):
    """The n-shard path of cilqr_multi_solve_batch (cilqr_shard_range, per-device host threads, the gather, the pick) with
    n = 2, 3, 8 shards on device 0 — ragged batches, batches smaller than the shard count (empty shards) — bit-equal to one
    handle.  (Shards sharing a device exchange their records by device copies; distinct devices use RCCL: the test above.)"""
    exe = _build_multi_device(tmp_path)
    for shards, B in ((2, 37), (3, 37), (8, 5), (3, 2), (2, 1), (8, 1100), (3, 1500)):
        out = _run_json([exe, "--shards", str(shards), "--B", str(B)])
        assert out["bit_equal"] and out["cases"] == 3 and out["rccl_cases"] == 0, out


def test_cpp_host_failure_with_copies_in_flight_leaves_nothing_behind(cilqr, tmp_path):
    """cilqr_debug_fail_enqueue on one shard's handle: the call fails AFTER other shards' (and its own) asynchronous copies were
    enqueued; it must report the error, drain every stream, and the next call on the same handles must succeed bit-equal."""
    exe = _build_multi_device(tmp_path)
    for shards, d, extra in ((3, 1, []), (2, 0, []), (3, 2, ["--pinned"]), (2, 1, ["--B", "1500"])):
        out = _run_json([exe, "--shards", str(shards), "--fail", str(d)] + extra)
        assert out["bit_equal"] and out["forced_failures"] == 1 and out["cases"] == 2, out
    out = _run_json([exe, "--fail", "0"])  # the one-device RCCL configuration
    assert out["bit_equal"] and out["forced_failures"] >= 1, out


def test_cpp_host_one_process_per_gpu_route(cilqr, tmp_path):
    """The other host model from a C++ process: RANK / WORLD_SIZE in the environment, RCCL id through a file
    (cilqr_comm_unique_id → cilqr_comm_init_rank), the shard by cilqr_shard_range, cilqr_argmin_global_device.  World size =
    the devices present (1 on the test box; the same command line runs per rank on a node)."""
    import os
    import subprocess
    exe = _build_multi_device(tmp_path)
    n = cilqr.lib().cilqr_device_count()
    idfile = str(tmp_path / "comm_id")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), CILQR_ID_FILE=idfile)
        procs.append(subprocess.Popen([exe, "--B", "41"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    import json
    for pr in procs:
        so, se = pr.communicate(timeout=300)
        lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
        assert pr.returncode == 0 and lines, so + se
        out = json.loads(lines[-1])
        assert out["bit_equal"] and out["world"] == n and out["best_single"] == out["best_global"], out


def test_host_call_failure_drains_and_frees_the_handle(cilqr, oracle):
    """cilqr_solve_batch with a forced failure after its input copies were enqueued (pinned buffers: true asynchronous DMA):
    error reported, the handle usable at once, the next call's results equal the oracle's."""
    from cilqr_amd import scenes
    N, M, B = 50, 4, 1200  # beyond the 1 MiB staging buffer: array-by-array copies straight from the caller's memory
    p = cilqr.default_params(N)
    sc = scenes.make_c2(B, p)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M)
    pin = {k: cilqr.pinned_copy(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")}
    out = dict(U=cilqr.pinned_empty((B, 2 * N)), X=cilqr.pinned_empty((B, 4 * (N + 1))), J=cilqr.pinned_empty((B,)),
               iters=cilqr.pinned_empty((B,), np.int32), status=cilqr.pinned_empty((B,), np.int32))
    out["U"][...] = sc["U"]
    cilqr._check(cilqr.lib().cilqr_debug_fail_enqueue(s._h, 1))
    with pytest.raises(cilqr.CilqrError) as e:
        s.solve_batch(N, pin["x0"], pin["U"], pin["poly"], pin["xplan_fl"], pin["obs_pose"], pin["obs_dim"], out=out)
    assert "forced failure" in str(e.value)
    out["U"][...] = sc["U"]
    got = s.solve_batch(N, pin["x0"], pin["U"], pin["poly"], pin["xplan_fl"], pin["obs_pose"], pin["obs_dim"], out=out)
    ns = 64
    want = oracle.solve_batch(oracle.default_params(N), N, M, *[np.ascontiguousarray(sc[k][:ns]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")], None, threads=16)
    assert np.max(np.abs(got["U"][:ns] - want["U"])) <= 1e-9 and np.array_equal(got["iters"][:ns], want["iters"])
    s.close()


# ---- costmap-lookup uncertainty cost (semantics defined by include/cilqr.h, PARITY UNPINNED) -------------------
def _unc_layer(oracle, seed, geom=(30.0, 20.0, 0.2, 15.0, 0.0)):
    from cilqr_amd import scenes
    og = oracle.map_geom(*geom)
    occ = scenes.make_occupancy(og.rows, og.cols, 90 + seed)
    layer, _, _ = oracle.blur(np.nan_to_num(occ, nan=0.0), og, np.sin(0.1), np.cos(0.1), 0.16, 0.16, 0.017, threads=16)
    layer[np.isnan(occ)] = np.nan
    return geom, layer


def _unc_params(mod, N):
    p = mod.default_params(N)
    p.safe_length, p.safe_width = 1.1, 0.9  # ilqr/launch/Experiment.launch:7-8
    return p


def test_uncertainty_cost_kernel_vs_oracle(cilqr, oracle):
    """The map cost alone (cilqr_debug_uncertainty_cost) against the plain-C statement of the same definition: random
    states on, beside and off the map, unknown cells, several probe grids and map poses.  Tolerance 1e-11 relative (the
    kernel's exp is within 2 ulp of libm's)."""
    geom, layer = _unc_layer(oracle, 0)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    rng = np.random.default_rng(11)
    for pose, probes in (((0.0, 0.0, 0.0), (3, 3)), ((2.0, -1.0, 0.3), (1, 1)), ((-40.0, 12.5, -2.1), (5, 2)), ((1.0, 1.0, 3.0), (2, 7))):
        p, po = _unc_params(cilqr, 50), _unc_params(oracle, 50)
        s = cilqr.Solver(p, max_batch=1, max_horizon=8, max_obstacles=0, device=0)
        try:
            s.set_uncertainty_map(layer, g, pose, probes)
            q = np.stack([rng.uniform(-3, 33, 512), rng.uniform(-12, 12, 512)], 1)
            c, sn = np.cos(pose[2]), np.sin(pose[2])
            st = np.stack([pose[0] + c * q[:, 0] - sn * q[:, 1], pose[1] + sn * q[:, 0] + c * q[:, 1], rng.uniform(0, 8, 512),
                           rng.uniform(-3.2, 3.2, 512)], 1)
            cost, vx, mx = s.debug_uncertainty_cost(st)
        finally:
            s.close()
        um, keep = oracle.uncertainty_map(layer, og, pose, probes)
        wc, wv, wm = oracle.uncertainty_cost(po, um, st)
        assert (wc > 0).mean() > 0.5 and (wc == 0).any()
        assert np.allclose(cost, wc, rtol=1e-11, atol=1e-14)
        assert np.allclose(vx, wv[:, :2], rtol=1e-11, atol=1e-12)
        assert np.allclose(mx, np.stack([wm[:, 0, 0], wm[:, 0, 1], wm[:, 1, 1]], 1), rtol=1e-11, atol=1e-12)


def test_uncertainty_cost_kernel_gradient_by_finite_differences(cilqr, oracle):
    """Self-consistency of the map cost on the device, independent of the formula it shares with its plain-C statement (ADVICE
    r02): vx must be the derivative of the kernel's OWN cost value with respect to (x, y) at fixed heading (central differences
    of cilqr_debug_uncertainty_cost), and with one probe mx must be the Gauss-Newton form vx vx' / x of that value.  A probe
    within the step of a cell-centre line sees the interpolant's kink: those few states are excluded, as in the oracle's test."""
    geom, layer = _unc_layer(oracle, 0)
    g = cilqr.map_geom(*geom)
    p = _unc_params(cilqr, 50)
    rng = np.random.default_rng(23)
    pose = (2.0, -1.0, 0.3)
    q = np.stack([rng.uniform(2, 28, 256), rng.uniform(-8, 8, 256)], 1)
    c, sn = np.cos(pose[2]), np.sin(pose[2])
    st = np.stack([pose[0] + c * q[:, 0] - sn * q[:, 1], pose[1] + sn * q[:, 0] + c * q[:, 1], np.full(256, 3.0), rng.uniform(-0.5, 0.5, 256)], 1)
    s = cilqr.Solver(p, max_batch=1, max_horizon=8, max_obstacles=0, device=0)
    try:
        s.set_uncertainty_map(layer, g, pose, (3, 3))
        cost, vx, mx = s.debug_uncertainty_cost(st)
        h = 1e-6
        for k in range(2):
            d = np.zeros(4)
            d[k] = h
            cp, _, _ = s.debug_uncertainty_cost(st + d)
            cm, _, _ = s.debug_uncertainty_cost(st - d)
            fd = (cp - cm) / (2 * h)
            smooth = np.abs(fd - vx[:, k]) < 1e-4 * (1 + np.abs(vx[:, k]))
            assert smooth.mean() > 0.9, (k, smooth.mean())
        assert (np.abs(vx).max(axis=1) > 1e-3).sum() > 20  # the scene does exercise the gradient
        s.set_uncertainty_map(layer, g, pose, (1, 1))
        c1, v1, m1 = s.debug_uncertainty_cost(st)
    finally:
        s.close()
    on = c1 > 0
    assert on.sum() > 100
    gn = np.stack([v1[:, 0] * v1[:, 0], v1[:, 0] * v1[:, 1], v1[:, 1] * v1[:, 1]], 1)[on] / c1[on, None]
    assert np.allclose(m1[on], gn, rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("G", [0, 8])
def test_solve_with_uncertainty_map_matches_oracle(cilqr, oracle, G, monkeypatch):
    """Whole solves with the map set (iLQR::set_uncertainty_map → w_uncertainty·(vx, mx) into l_x, l_xx at every step,
    I/Constraints.cpp:188-201), both kernel families, shared map: U within 1e-9 of the oracle, iterations and exits equal;
    clearing the map restores the plain solve bit for bit."""
    from cilqr_amd import scenes
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    N, M, B = 50, 4, 192
    p, po = _unc_params(cilqr, N), _unc_params(oracle, N)
    sc = scenes.make_static(B, N, M, p, 4311)
    geom, layer = _unc_layer(oracle, 1)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    pose = (-1.0, 0.4, 0.05)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        plain = _gpu_batch(s, sc)
        s.set_uncertainty_map(layer, g, pose, (3, 3))
        got = _gpu_batch(s, sc)
        s.clear_uncertainty_map()
        plain2 = _gpu_batch(s, sc)
    finally:
        s.close()
    um, keep = oracle.uncertainty_map(layer, og, pose, (3, 3))
    want = oracle.solve_batch_unc(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, um,
                                  threads=min(16, oracle.max_threads()))
    _compare(got, want, TIGHT, "uncertainty map G=%d" % G)
    for k in ("U", "X", "J", "iters", "status"):
        assert np.array_equal(plain[k], plain2[k]), k
    assert np.abs(got["U"] - plain["U"]).max() > 1e-3  # the map term is live


@pytest.mark.parametrize("N,M,B,W", [(50, 4, 192, 3), (50, 4, 700, 2), (30, 0, 40, 3), (80, 6, 100, 2), (50, 4, 1500, 1)])
def test_share_kernel_with_uncertainty_map_changes_no_bit(cilqr, oracle, monkeypatch, N, M, B, W):
    """With a map set the shared-phase-L kernel gives the map term to its last aux wavefront (three wavefronts per solve up to half a
    solve per SIMD, two up to one, one beyond; horizons beyond 64 on two): U, X, J, iterations and exits bit-identical to the
    one-wavefront kernel's with the same map, a shared map and per-solve maps with per-solve poses."""
    import torch
    from cilqr_amd import scenes
    p = _unc_params(cilqr, N)
    sc = scenes.make_static(B, N, M, p, 4411)
    geom, layer = _unc_layer(oracle, 2)
    g = cilqr.map_geom(*geom)
    rng = np.random.default_rng(4412)
    poses = np.stack([rng.uniform(-2, 2, B), rng.uniform(-1, 1, B), rng.uniform(-0.2, 0.2, B)], 1)
    dev = torch.device("cuda", 0)
    d_layer = torch.from_numpy(np.ascontiguousarray(np.asfortranarray(layer, dtype=np.float32).flatten(order="F"))).to(dev)
    d_poses = torch.from_numpy(poses).to(dev)
    out = {}
    for name in ("share", "one"):
        if name == "one":
            monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")
        s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
        monkeypatch.delenv("CILQR_NO_SHARE_KERNEL", raising=False)
        try:
            s.set_uncertainty_map(layer, g, (-1.0, 0.4, 0.05), (3, 3))
            w = s.solve_wavefronts(B, N, M)
            shared = _gpu_batch(s, sc)
            s.set_uncertainty_map_device(d_layer.data_ptr(), g, (0.0, 0.0, 0.0), (2, 3), layer_stride=0, poses_ptr=d_poses.data_ptr())
            per_pose = _gpu_batch(s, sc)
        finally:
            s.close()
        out[name] = (w, shared, per_pose)
    assert (out["share"][0], out["one"][0]) == (W, 1)
    _same_bits(out["share"][1], out["one"][1], "map set, %d wavefronts against one" % W)
    _same_bits(out["share"][2], out["one"][2], "per-solve poses, %d wavefronts against one" % W)
    assert np.isfinite(out["share"][1]["U"]).all()


def test_solve_with_per_solve_maps_and_poses(cilqr, oracle):
    """The device form with one layer and one map pose per solve (a scenario batch), wavefront family and sampled obstacles
    on top: every solve reads its own map."""
    import torch
    from cilqr_amd import scenes
    N, B = 50, 48
    p, po = _unc_params(cilqr, N), _unc_params(oracle, N)
    sc = scenes.make_c3(B, p, n_dyn=3, n_samples=4)
    geom = (30.0, 20.0, 0.2, 15.0, 0.0)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    layers = np.stack([_unc_layer(oracle, 10 + (b % 4))[1] for b in range(B)])
    rng = np.random.default_rng(23)
    poses = np.stack([rng.uniform(-2, 1, B), rng.uniform(-1, 1, B), rng.uniform(-0.1, 0.1, B)], 1)
    dev = torch.device("cuda", 0)
    flat = np.ascontiguousarray(np.stack([np.asfortranarray(a).flatten(order="F") for a in layers]))
    d_layers = torch.from_numpy(flat).to(dev)
    d_poses = torch.from_numpy(poses).to(dev)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=sc["M"], device=0)
    try:
        s.set_uncertainty_map_device(d_layers.data_ptr(), g, (0.0, 0.0, 0.0), (2, 3), layer_stride=flat.shape[1], poses_ptr=d_poses.data_ptr())
        got = s.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["nom_pose"], sc["nom_dim"], sc["offsets"],
                                    sc["sample_weight"])
        gotm = s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"])
    finally:
        s.close()
    um, keep = oracle.uncertainty_map(layers, og, (0.0, 0.0, 0.0), (2, 3), poses=poses, batched=True)
    want = oracle.solve_batch_unc(po, N, sc["M"], sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"],
                                  sc["obs_weight"], um, threads=min(16, oracle.max_threads()))
    _compare(gotm, want, TIGHT, "per-solve maps, materialised obstacles")
    _compare(got, want, 1e-8, "per-solve maps, sampled obstacles")


def test_frame_then_solve_on_one_stream(cilqr, oracle):
    """The deployed sequence on ONE stream with nothing leaving the device in between: map node's frame
    (cilqr_costmap_frame_device: warp → blur) → its uncertainty layer set as the planner's map → batched solve.  The oracle
    is given the layer the device produced (the frame's own parity is test_costmap_frame_equals_its_three_steps)."""
    import torch
    from cilqr_amd import scenes
    N, M, B = 50, 2, 64
    p, po = _unc_params(cilqr, N), _unc_params(oracle, N)
    sc = scenes.make_static(B, N, M, p, 977)
    sgeo, vgeo = (120.0, 120.0, 0.2, 10.0, 0.0), (30.0, 20.0, 0.2, 15.0, 0.0)
    sg, vg, ovg = cilqr.map_geom(*sgeo), cilqr.map_geom(*vgeo), oracle.map_geom(*vgeo)
    glob = scenes.make_occupancy(sg.rows, sg.cols, 5, n_blobs=250, nan_frac=0.0)
    pose = (0.5, -0.3, 0.08)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    d_glob = torch.from_numpy(np.ascontiguousarray(glob.T)).to(dev)
    veh = torch.zeros(vg.rows * vg.cols, dtype=torch.float32, device=dev)
    unc = torch.zeros_like(veh)
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim")}
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev)
    J = torch.zeros(B, dtype=torch.float64, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        s.costmap_frame_device(side.cuda_stream, d_glob.data_ptr(), sg, vg, *pose, 0.16, 0.16, 0.017, veh.data_ptr(), unc.data_ptr())
        s.set_uncertainty_map_device(unc.data_ptr(), vg, pose, (3, 3))
        s.solve_batch_device(side.cuda_stream, B, N, M, t["x0"].data_ptr(), t["U"].data_ptr(), t["poly"].data_ptr(), t["xplan_fl"].data_ptr(),
                             t["obs_pose"].data_ptr(), t["obs_dim"].data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        side.synchronize()
    finally:
        s.close()
    layer = unc.cpu().numpy().reshape(vg.cols, vg.rows).T
    assert np.nanmax(layer) > 50  # the frame did put obstacles under the planner
    um, keep = oracle.uncertainty_map(layer, ovg, pose, (3, 3))
    want = oracle.solve_batch_unc(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, um,
                                  threads=min(16, oracle.max_threads()))
    got = dict(U=t["U"].cpu().numpy(), X=X.cpu().numpy(), J=J.cpu().numpy(), iters=it.cpu().numpy(), status=st.cpu().numpy())
    _compare(got, want, TIGHT, "frame -> solve")


def test_cpp_adapter_uncertainty_map(cilqr, oracle, tmp_path):
    """iLQR::set_uncertainty_map / clear_uncertainty_map of the C++ façade, in the reference node's order (set map, set plan,
    run_step; I/ilqr_uncertainty_node.cpp:111-119), against the oracle."""
    import json
    import os
    import subprocess
    from conftest import PKG, ROOT
    from cilqr_amd import scenes
    exe = str(tmp_path / "adapter_uncertainty")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "adapter_uncertainty.cpp"), "-L" + os.path.join(PKG, "lib"), "-lcilqr_hip",
                    "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True)
    geom, layer = _unc_layer(oracle, 2)
    og = oracle.map_geom(*geom)
    path = tmp_path / "layer.bin"
    np.asfortranarray(layer).flatten(order="F").astype(np.float32).tofile(path)
    pose = (-0.5, 0.2, 0.03)
    r = subprocess.run([exe, str(path), str(og.rows), str(og.cols)] + [repr(v) for v in geom] + [repr(v) for v in pose],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    N, M = 50, 4
    po = _unc_params(oracle, N)
    sc = scenes.known_answer_scene(N, M, po, local_plan=oracle.local_plan)
    um, keep = oracle.uncertainty_map(layer, og, pose, (3, 3))
    w_on = oracle.solve_batch_unc(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, um)
    w_off = oracle.solve_batch_unc(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, None)
    assert out["with_map"]["iterations"] == int(w_on["iters"][0]) and out["cleared"]["iterations"] == int(w_off["iters"][0])
    assert np.max(np.abs(np.array(out["with_map"]["U"]) - w_on["U"][0])) < TIGHT
    assert np.max(np.abs(np.array(out["cleared"]["U"]) - w_off["U"][0])) < TIGHT
    assert np.max(np.abs(w_on["U"] - w_off["U"])) > 1e-3


@pytest.mark.parametrize("variant", ["default", "lds_tiles", "eight_rows"])
def test_warp_batch_equals_single_frames(cilqr, oracle, monkeypatch, variant):
    """cilqr_warp_costmap_batch_device (K frames per launch, 16-byte stores) against the single-frame kernel and the oracle,
    bit for bit: config-4 shape with out-of-range frames and a bbox layer, a small map whose rows are a multiple of 4 but
    not of the tile, and a map whose rows are not a multiple of 4 (frame-by-frame fallback).  Also the two kernels kept as
    measured-and-dropped experiments: the LDS-tiled one (CILQR_WARP_LDS=1) and eight rows per lane (CILQR_WARP_ROWS=8)."""
    import torch
    if variant == "lds_tiles":
        monkeypatch.setenv("CILQR_WARP_LDS", "1")
    if variant == "eight_rows":
        monkeypatch.setenv("CILQR_WARP_ROWS", "8")
    from cilqr_amd import scenes
    dev = torch.device("cuda", 0)
    s = cilqr.Solver(cilqr.default_params(), max_batch=1, max_horizon=1, max_obstacles=0, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        c4 = scenes.make_c4()
        cases = [(c4["src"], c4["src_geom"], c4["dst_geom"], np.concatenate([c4["poses"][::43], [[95.0, 0.0, 0.4], [0.0, -70.0, 2.0]]]), True),
                 (c4["src"][:300, :260], (60.0, 52.0, 0.2, 3.0, -2.0), (10.0, 7.4, 0.1, 5.0, 0.0), c4["poses"][5:9] * [0.3, 0.3, 1.0], False),
                 (c4["src"][:300, :260], (60.0, 52.0, 0.2, 3.0, -2.0), (15.0, 10.0, 0.1, 7.5, 0.0), c4["poses"][17:20] * [0.3, 0.3, 1.0], True)]
        for src_h, sgeo, dgeo, poses, with_bbox in cases:
            src_h = np.asfortranarray(src_h)
            sg, dg, osg, odg = cilqr.map_geom(*sgeo), cilqr.map_geom(*dgeo), oracle.map_geom(*sgeo), oracle.map_geom(*dgeo)
            K, cells = len(poses), dg.rows * dg.cols
            rng = np.random.default_rng(K)
            bbox_h = None
            if with_bbox:
                bbox_h = np.zeros((dg.rows, dg.cols), dtype=np.float32)
                bbox_h[rng.random(bbox_h.shape) < 0.03] = 100.0
            d_src = torch.from_numpy(np.ascontiguousarray(src_h.T)).to(dev)
            d_bbox = torch.from_numpy(np.ascontiguousarray(bbox_h.T)).to(dev) if with_bbox else None
            d_dst = torch.zeros(K * cells, dtype=torch.float32, device=dev)
            d_oob = torch.full((K,), -1, dtype=torch.int64, device=dev)
            s.warp_costmap_batch_device(stream, d_src.data_ptr(), sg, d_dst.data_ptr(), dg, poses, d_bbox.data_ptr() if with_bbox else 0,
                                        d_oob.data_ptr())
            torch.cuda.synchronize()
            got = d_dst.cpu().numpy().reshape(K, dg.cols, dg.rows)
            for k in range(K):
                want, woob = oracle.warp(src_h, osg, odg, *poses[k], bbox=bbox_h, threads=16)
                assert np.array_equal(got[k].T, want, equal_nan=True), (dgeo, k)
                assert int(d_oob[k].item()) == woob
                single, soob = s.warp_costmap(src_h, sg, dg, *poses[k], bbox=bbox_h)
                assert np.array_equal(single, want, equal_nan=True) and soob == woob
    finally:
        s.close()


def test_sampled_obstacles_with_changing_shape(cilqr, oracle):
    """The compact sampled form derives a sample's heading and semi-axes once per solve when its obstacle keeps heading, speed
    and dimensions over the horizon, and per entry otherwise.  Here one obstacle turns, one brakes, one grows and one is
    steady — in the same solve — against the oracle on the materialised scene (1e-8: angle addition, reciprocal semi-axes)."""
    from cilqr_amd import scenes
    N, B, n_dyn, S = 50, 40, 4, 6
    p = cilqr.default_params(N)
    sc = scenes.make_c3(B, p, n_dyn=n_dyn, n_samples=S)
    nom = sc["nom_pose"].reshape(B, n_dyn, N, 4).copy()
    dim = sc["nom_dim"].reshape(B, n_dyn, N, 2).copy()
    t = np.arange(N)
    nom[:, 0, :, 3] += 0.01 * t                       # turning
    nom[:, 1, :, 2] *= np.linspace(1.0, 0.3, N)       # braking
    dim[:, 2, N // 2:, 0] += 0.5                      # grows half-way
    off = sc["offsets"]
    poses = np.repeat(nom[:, :, None, :, :], S, axis=2)
    poses[..., 0] += off[..., 0][..., None]
    poses[..., 1] += off[..., 1][..., None]
    poses[..., 3] += off[..., 2][..., None]
    dims = np.repeat(dim[:, :, None, :, :], S, axis=2)
    M = n_dyn * S
    mat = dict(sc, M=M, obs_pose=poses.reshape(B, M, 4 * N), obs_dim=dims.reshape(B, M, 2 * N))
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=0)
    try:
        got = s.solve_batch_sampled(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], nom.reshape(B, n_dyn, 4 * N), dim.reshape(B, n_dyn, 2 * N),
                                    off, sc["sample_weight"])
    finally:
        s.close()
    _compare(got, _oracle_batch(oracle, N, mat), 1e-8, "sampled, changing shape")
