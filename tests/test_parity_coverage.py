"""Which oracle-only behaviours the parity sets actually reach.

The whole solve of the reference cannot be built in this image (every solver translation unit includes mkl.h and an
Uncertainty.h the reference does not ship), so four behaviours rest on the oracle's restatement being read correctly, with no
reference-run evidence behind them: the speed inflation of moving obstacles (I/Obstacle.cpp:42-43), the tolerance exit
(I/iLQR.cpp:227-230), warm starts (I/iLQR.cpp:253) and per-obstacle weights (I/Constraints.cpp:184-185).  This module does not
pin them — nothing here can — it makes visible how much of each parity set runs through them: for the scenes the `-m gpu` tests
compare against the oracle (config 2 in full, the config-3 and config-5 samples of the full-size tests) it tabulates the exit
reasons and the share of solves with moving obstacles, per-obstacle weights and a warm start.

    python tests/test_parity_coverage.py > profiles/rNN_parity_coverage.txt      # the same table, as a committed record
"""
import os
import sys

import numpy as np

EXITS = {0: "tolerance", 1: "lambda_max", 2: "max_iter", 3: "numeric"}


def parity_sets():
    """(name, N, scene dict) of the scenes the GPU parity tests hand to the oracle, generated exactly as they generate them."""
    import cilqr_amd
    from cilqr_amd import scenes
    p50, p80 = cilqr_amd.default_params(50), cilqr_amd.default_params(80)
    c2 = scenes.make_c2(1024, p50)  # test_config2_batch_1024: all 1024
    c3 = scenes.make_c3(4096, p50)  # test_config3_full_batch_properties: 128-solve sample of the full batch
    i3 = np.concatenate([np.arange(48), np.arange(2000, 2040), np.arange(4096 - 40, 4096)])
    c5 = scenes.make_c5(8192, p80)  # test_config5_shard_full_size_properties: 192-solve sample
    i5 = np.concatenate([np.arange(64), np.arange(4000, 4064), np.arange(8192 - 64, 8192)])

    def sub(sc, idx):
        return {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    return [("config 2 (B=1024, all)", 50, c2), ("config 3 (128-solve sample)", 50, sub(c3, i3)), ("config 5 (192-solve sample)", 80, sub(c5, i5))]


def coverage(O, N, sc):
    """Exit-reason histogram and oracle-only-path shares of one parity set."""
    p = O.default_params(N)
    r = O.solve_batch(p, N, sc["M"], sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], sc["obs_weight"],
                      threads=min(8, O.max_threads()))
    B, M = r["iters"].size, sc["M"]
    speed = sc["obs_pose"].reshape(B, M, N, 4)[..., 2]
    default_U = np.tile(np.array([[0.5, 0.0]] * (N // 2) + [[0.5, 0.1]] * (N - N // 2)).reshape(-1), (B, 1))  # I/iLQR.cpp:8-15
    return {"solves": B,
            "exits": {EXITS[k]: int(np.sum(r["status"] == k)) for k in EXITS},
            "iterations": (int(r["iters"].min()), float(r["iters"].mean()), int(r["iters"].max())),
            "moving_obstacles": float(np.mean(np.any(speed != 0.0, axis=(1, 2)))),
            "per_obstacle_weights": 0.0 if sc["obs_weight"] is None else 1.0,
            "warm_started": float(np.mean(np.any(sc["U"].reshape(B, -1) != default_U, axis=1)))}


def table(O):
    rows = []
    for name, N, sc in parity_sets():
        c = coverage(O, N, sc)
        rows.append((name, c))
    return rows


def test_parity_sets_reach_the_oracle_only_paths(oracle, capsys):
    """Prints the table (visible with -s or on failure) and asserts what the documentation says about it: config 3 is the set
    that exercises moving obstacles and per-obstacle weights; the tolerance exit is reached by a few solves of configs 2 and 3
    (22 of 1024, 6 of 128) and by none of config 5's sample; no full-size set starts warm — warm starts are covered by the
    dedicated GPU tests (warm-start cases, the adapter replay), against the same oracle."""
    rows = table(oracle)
    with capsys.disabled():
        print()
        for name, c in rows:
            print("%-30s %s" % (name, c))
    by = dict(rows)
    for name, c in rows:
        assert sum(c["exits"].values()) == c["solves"] and c["exits"]["numeric"] == 0, name
    assert by["config 2 (B=1024, all)"]["moving_obstacles"] == 0.0
    assert by["config 3 (128-solve sample)"]["moving_obstacles"] == 1.0
    assert by["config 3 (128-solve sample)"]["per_obstacle_weights"] == 1.0
    assert by["config 5 (192-solve sample)"]["moving_obstacles"] == 0.0
    assert by["config 2 (B=1024, all)"]["exits"]["tolerance"] > 0 and by["config 3 (128-solve sample)"]["exits"]["tolerance"] > 0
    assert all(c["warm_started"] == 0.0 for _, c in rows)


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for q in (ROOT, os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")):
        sys.path.insert(0, q)
    from oracle import oracle as O
    O.build(ref=False)
    print("# parity sets against the oracle: exit reasons and the share of solves on oracle-only paths (tests/test_parity_coverage.py)")
    for name, c in table(O):
        print("%s: %d solves; exits %s; reference iterations min/mean/max %d / %.2f / %d; moving obstacles %.0f %%; "
              "per-obstacle weights %.0f %%; warm-started %.0f %%"
              % (name, c["solves"], c["exits"], c["iterations"][0], c["iterations"][1], c["iterations"][2],
                 100 * c["moving_obstacles"], 100 * c["per_obstacle_weights"], 100 * c["warm_started"]))
