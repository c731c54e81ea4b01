"""CPU tests of the seeded scene generators (host-side numpy; no GPU): the closed-loop tick sequence that bench.py and
tools/schedule_order_ticks.py replay starts at the benchmark batch and moves on the way the reference node does between two
run_step calls (ego to the plan's first state, controls kept un-shifted, obstacles moved on by one timestep, fresh pose noise)."""
import numpy as np


def test_tick_sequence_starts_at_the_benchmark_batch_and_advances(cilqr, oracle):
    from cilqr_amd import scenes
    N = 50
    p = cilqr.default_params(N)
    for kind, ref in (("c3", scenes.make_c3(12, p, oracle.local_plan, 3, 4)), ("static", scenes.make_static(12, N, 4, p, scenes.SEED0 + 2, oracle.local_plan))):
        ts = scenes.TickSequence(kind, 12, p, N=N, M=4, local_plan=oracle.local_plan, n_dyn=3, n_samples=4)
        i0 = ts.inputs()
        keys = ("x0", "U", "poly", "xplan_fl") + (("nom_pose", "nom_dim", "offsets") if kind == "c3" else ("obs_pose", "obs_dim"))
        for k in keys:
            assert np.array_equal(i0[k], ref[k]), (kind, k)
        # a made-up result: the plan's first state and some controls
        rng = np.random.default_rng(1)
        X = np.zeros((12, N + 1, 4))
        X[:, 0] = i0["x0"]
        X[:, 1] = i0["x0"] + np.array([0.3, 0.01, 0.05, 0.002])
        U = rng.normal(0.0, 0.3, (12, 2 * N))
        ts.advance(X.reshape(12, -1), U)
        i1 = ts.inputs()
        assert i1["tick"] == 1
        assert np.array_equal(i1["x0"], X[:, 1]) and np.array_equal(i1["U"], U)  # warm start, un-shifted (I/iLQR.cpp:253)
        pk = "nom_pose" if kind == "c3" else "obs_pose"
        a, b = i0[pk].reshape(12, -1, N, 4), i1[pk].reshape(12, -1, N, 4)
        # every obstacle is where it would have been one timestep later: column t of the new tick = column t + 1 of the old one
        assert np.allclose(b[:, :, :-1, :2], a[:, :, 1:, :2], rtol=0, atol=1e-12)
        assert np.array_equal(b[..., 2:], a[..., 2:])
        if kind == "c3":
            assert not np.array_equal(i1["offsets"], i0["offsets"]) and i1["offsets"].shape == i0["offsets"].shape
            m = scenes.materialise_samples(ref["nom_pose"], ref["nom_dim"], ref["offsets"], N)
            assert np.array_equal(m[0], ref["obs_pose"]) and np.array_equal(m[1], ref["obs_dim"]) and np.array_equal(m[2], ref["obs_weight"])
        # the local plan was re-fitted around the new ego (the oracle's LocalPlanner, pinned to the reference's Eigen)
        A, om, ph = ts.curve
        xs = np.arange(200.0)
        c, r = oracle.local_plan(p, np.stack([xs, A[0] * np.sin(om[0] * xs + ph[0])], 1), i1["x0"][0])
        assert np.array_equal(i1["poly"][0], c) and i1["xplan_fl"][0, 0] == r[0, 0] and i1["xplan_fl"][0, 1] == r[-1, 0]
