"""Randomised parity sweep (`-m gpu`): shapes, kernel families and lane groupings, parameter sets, weights and warm starts drawn
from a seeded generator, every draw through the C-ABI against the oracle.  The fixed cases of test_gpu_parity.py pin named
behaviours; this sweep looks for what nobody thought of naming — in particular in the paths added in round 3 (lane sharing in
the grouped family, two and four wavefronts per sampled solve, two and three per static-obstacle solve), whose lane ↔ step maps
depend on the shape in many ways."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# CILQR_FUZZ_SCALE=n in the environment: n times as many seeded draws per test (a one-off wider sweep; the committed default is 1)
SCALE = max(1, int(os.environ.get("CILQR_FUZZ_SCALE", "1")))

TIGHT = 1e-9


def _compare(got, want, what):
    same = (got["iters"] == want["iters"]) & (got["status"] == want["status"])
    assert same.all(), "%s: %d solves took a different accept/reject path" % (what, int((~same).sum()))
    fin = np.isfinite(want["U"]).all(axis=1)
    assert np.array_equal(np.isfinite(got["U"]).all(axis=1), fin), what
    if fin.any():
        assert np.max(np.abs(got["U"][fin] - want["U"][fin])) <= TIGHT, what
        assert np.max(np.abs(got["X"][fin] - want["X"][fin])) <= 100 * TIGHT, what
        assert np.allclose(got["J"][fin], want["J"][fin], rtol=1e-9, atol=1e-9), what


def _tweak(p, rng):
    """A parameter set off the defaults (both bindings' structs carry the same field names)."""
    p.w_pos = float(rng.uniform(0.3, 1.5)); p.w_vel = float(rng.uniform(1.0, 5.0)); p.w_acc = float(rng.uniform(0.5, 2.0))
    p.w_yawrate = float(rng.uniform(2.0, 6.0)); p.desired_speed = float(rng.uniform(3.0, 8.0)); p.w_obstacle = float(rng.uniform(0.5, 2.0))
    p.q2_front = float(rng.uniform(2.0, 3.5)); p.q2_rear = float(rng.uniform(2.0, 3.0)); p.t_safe = float(rng.uniform(0.0, 0.3))
    p.max_iterations = int(rng.integers(3, 21)); p.tolerance = float(10.0 ** rng.uniform(-5, -2))
    return p


@pytest.mark.parametrize("seed", range(18 * SCALE))
def test_random_static_scenes(cilqr, oracle, monkeypatch, seed):
    from cilqr_amd import scenes
    rng = np.random.default_rng(9100 + seed)
    N, M, B = int(rng.integers(2, 91)), int(rng.integers(0, 13)), int(rng.integers(1, 301))
    G = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
    if G:
        monkeypatch.setenv("CILQR_FORCE_G", str(G))
    # one wavefront per solve, or two / three sharing phase L where that kernel applies (a generator of its own: the draws above and
    # below are those of the first version of this test)
    share = str(np.random.default_rng(9500 + seed).choice(["rule", "2", "3", "off"]))
    if share == "off":
        monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")
    elif share != "rule":
        monkeypatch.setenv("CILQR_SHARE_W", share)
    p, po = cilqr.default_params(N), oracle.default_params(N)
    if seed % 2:
        state = rng.bit_generator.state
        _tweak(p, rng)
        rng.bit_generator.state = state
        _tweak(po, rng)
    sc = scenes.make_static(B, N, M, p, 9200 + seed)
    if M and rng.random() < 0.5:
        sc["obs_weight"] = rng.uniform(0.2, 2.0, (B, M))
    if rng.random() < 0.5:
        sc["U"] = sc["U"] + rng.normal(0.0, 0.25, sc["U"].shape)
    if M and rng.random() < 0.5:  # moving obstacles
        pose = sc["obs_pose"].reshape(B, M, N, 4).copy()
        v = rng.uniform(0.0, 6.0, (B, M, 1))
        pose[..., 2] = v
        pose[..., 0] += v * np.cos(pose[..., 3]) * p.timestep * np.arange(N)
        pose[..., 1] += v * np.sin(pose[..., 3]) * p.timestep * np.arange(N)
        sc["obs_pose"] = pose.reshape(B, M, 4 * N)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    try:
        got = s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"])
    finally:
        s.close()
    want = oracle.solve_batch(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"],
                              threads=min(16, oracle.max_threads()))
    _compare(got, want, "seed %d: N=%d M=%d B=%d G=%d share=%s" % (seed, N, M, B, G, share))


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_random_static_scenes_wavefront_family(cilqr, oracle, monkeypatch, seed):
    """The one-wavefront family on its own: one, two or three wavefronts per solve (cilqr_solve_share_kernel where the shape allows it,
    cilqr_solve_kernel otherwise), horizons up to the 63 the shared kernel takes and a little beyond, tables that fit LDS and tables
    that do not, weights, warm starts, moving obstacles, parameter sets off the defaults."""
    from cilqr_amd import scenes
    rng = np.random.default_rng(9600 + seed)
    N, M, B = int(rng.integers(1, 70)), int(rng.integers(0, 11)), int(rng.integers(1, 700))
    if seed >= 8:  # (added with the long-horizon form of the shared kernel: horizons up to 130)
        N += 62
    share = ["rule", "2", "3", "off"][seed % 4]
    monkeypatch.setenv("CILQR_FORCE_G", "64")
    if share == "off":
        monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")
    elif share != "rule":
        monkeypatch.setenv("CILQR_SHARE_W", share)
    p, po = cilqr.default_params(N), oracle.default_params(N)
    if seed % 3 == 0:
        state = rng.bit_generator.state
        _tweak(p, rng)
        rng.bit_generator.state = state
        _tweak(po, rng)
    sc = scenes.make_static(B, N, M, p, 9700 + seed)
    if M and rng.random() < 0.5:
        sc["obs_weight"] = rng.uniform(0.2, 2.0, (B, M))
    if rng.random() < 0.5:
        sc["U"] = sc["U"] + rng.normal(0.0, 0.25, sc["U"].shape)
    if M and rng.random() < 0.5:
        pose = sc["obs_pose"].reshape(B, M, N, 4).copy()
        v = rng.uniform(0.0, 6.0, (B, M, 1))
        pose[..., 2] = v
        pose[..., 0] += v * np.cos(pose[..., 3]) * p.timestep * np.arange(N)
        pose[..., 1] += v * np.sin(pose[..., 3]) * p.timestep * np.arange(N)
        sc["obs_pose"] = pose.reshape(B, M, 4 * N)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    try:
        w = s.solve_wavefronts(B, N, M)
        got = s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"])
    finally:
        s.close()
    want = oracle.solve_batch(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"],
                              threads=min(16, oracle.max_threads()))
    _compare(got, want, "seed %d: N=%d M=%d B=%d share=%s -> %d wavefront(s) per solve" % (seed, N, M, B, share, w))


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_static_scenes_with_map(cilqr, oracle, monkeypatch, seed):
    """The same with an uncertainty map set (the reference's own mode: w_uncertainty·(vx, mx) into l_x, l_xx at every step,
    I/Constraints.cpp:188-201): random probe grids and map poses, horizons up to 127, one to three wavefronts per solve (the map term
    on the last aux wavefront of cilqr_solve_share_kernel) and the grouped family."""
    from cilqr_amd import scenes
    rng = np.random.default_rng(9800 + seed)
    N, M, B = int(rng.integers(2, 128)), int(rng.integers(0, 9)), int(rng.integers(1, 400))
    mode = ["rule", "2", "off", "g8"][seed % 4]
    if mode == "g8":
        monkeypatch.setenv("CILQR_FORCE_G", "8")
    else:
        monkeypatch.setenv("CILQR_FORCE_G", "64")
        if mode == "off":
            monkeypatch.setenv("CILQR_NO_SHARE_KERNEL", "1")
        elif mode != "rule":
            monkeypatch.setenv("CILQR_SHARE_W", mode)
    p, po = cilqr.default_params(N), oracle.default_params(N)
    for q in (p, po):
        q.safe_length, q.safe_width = 1.1, 0.9  # ilqr/launch/Experiment.launch:7-8
    sc = scenes.make_static(B, N, M, p, 9900 + seed)
    if rng.random() < 0.5:
        sc["U"] = sc["U"] + rng.normal(0.0, 0.1, sc["U"].shape)
    geom = (30.0, 20.0, 0.2, 15.0, 0.0)
    g, og = cilqr.map_geom(*geom), oracle.map_geom(*geom)
    occ = scenes.make_occupancy(og.rows, og.cols, 700 + seed)
    layer, _, _ = oracle.blur(np.nan_to_num(occ, nan=0.0), og, np.sin(0.1), np.cos(0.1), 0.16, 0.16, 0.017, threads=16)
    layer[np.isnan(occ)] = np.nan
    pose = (float(rng.uniform(-3, 3)), float(rng.uniform(-1, 1)), float(rng.uniform(-0.3, 0.3)))
    probes = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=max(M, 1), device=0)
    try:
        s.set_uncertainty_map(layer, g, pose, probes)
        w = s.solve_wavefronts(B, N, M)
        got = s.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"])
    finally:
        s.close()
    um, keep = oracle.uncertainty_map(layer, og, pose, probes)
    want = oracle.solve_batch_unc(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"], um,
                                  threads=min(16, oracle.max_threads()))
    _compare(got, want, "seed %d: N=%d M=%d B=%d probes %s mode %s -> %d wavefront(s) per solve" % (seed, N, M, B, probes, mode, w))


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_sampled_scenes(cilqr, oracle, monkeypatch, seed):
    from cilqr_amd import scenes
    rng = np.random.default_rng(9300 + seed)
    N, n_dyn, S, B = int(rng.integers(3, 70)), int(rng.integers(1, 9)), int(rng.integers(2, 12)), int(rng.integers(1, 120))
    W = int(rng.choice([0, 2, 4]))
    if W:
        monkeypatch.setenv("CILQR_SPLIT_W", str(W))
    else:
        monkeypatch.setenv("CILQR_NO_SPLIT_KERNEL", "1")
    p, po = cilqr.default_params(N), oracle.default_params(N)
    st = scenes.make_static(B, N, n_dyn, p, 9400 + seed)
    pose = st["obs_pose"].reshape(B, n_dyn, N, 4).copy()
    v = rng.uniform(0.0, 7.0, (B, n_dyn, 1))
    pose[..., 2] = v
    pose[..., 0] += v * np.cos(pose[..., 3]) * p.timestep * np.arange(N)
    pose[..., 1] += v * np.sin(pose[..., 3]) * p.timestep * np.arange(N)
    if n_dyn > 1:
        pose[:, 1, :, 3] += 0.02 * np.arange(N)  # one obstacle turns: its samples are derived per entry
    off = rng.normal(0.0, 1.0, (B, n_dyn, S, 3)) * scenes.POSE_SIGMA
    nom_pose, nom_dim = pose.reshape(B, n_dyn, 4 * N), st["obs_dim"]
    mp, md, mw = scenes.materialise_samples(nom_pose, nom_dim, off, N)
    s = cilqr.Solver(p, max_batch=B, max_horizon=N, max_obstacles=n_dyn * S, device=0)
    try:
        got = s.solve_batch_sampled(N, st["x0"], st["U"], st["poly"], st["xplan_fl"], nom_pose, nom_dim, off, 1.0 / S)
    finally:
        s.close()
    want = oracle.solve_batch(po, N, n_dyn * S, st["x0"], st["U"], st["poly"], st["xplan_fl"], mp, md, mw, threads=min(16, oracle.max_threads()))
    _compare(got, want, "seed %d: N=%d n_dyn=%d S=%d B=%d W=%d" % (seed, N, n_dyn, S, B, W))
