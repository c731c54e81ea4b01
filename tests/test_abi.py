"""CPU tests of the product's C-ABI shared library: it loads, exports every symbol include/cilqr.h declares, its host-side
pieces (parameter defaults, warm-start sequence, LocalPlanner pre-step, map geometry) match the reference-pinned values,
and it has NO CPU compute path (create fails loudly without a gfx950 device).  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol(cilqr):
    L = cilqr.lib()
    header = open(os.path.join(ROOT, "include", "cilqr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(cilqr_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(cilqr.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.cilqr_abi_version() == 2


def test_struct_layouts_agree(cilqr, oracle):
    assert C.sizeof(cilqr.Params) == C.sizeof(oracle.Params) == 6 * 4 + 41 * 8
    assert C.sizeof(cilqr.MapGeom) == 2 * 4 + 5 * 8


def test_params_default_matches_reference(cilqr, oracle):
    mine = cilqr.default_params().as_dict()
    ref = load_golden("ref_params.json")["params"]
    for k, v in ref.items():
        assert mine[k] == v, k
    assert mine == oracle.default_params().as_dict()


def test_default_control_seq(cilqr, oracle):
    for N in (1, 7, 30, 50, 80):
        assert np.array_equal(cilqr.default_control_seq(N), oracle.default_control_seq(N))
    with pytest.raises(cilqr.CilqrError):
        cilqr.default_control_seq(0)


def test_local_plan_matches_reference_polyfit(cilqr, oracle):
    p = cilqr.default_params()
    for c in load_golden("ref_polyfit.json")["cases"]:
        if c["degree"] != 5 or len(c["x"]) > 20:
            continue
        x, y = np.array(c["x"]), np.array(c["y"])
        path = np.stack([x, y], 1)
        coeffs, ref = cilqr.local_plan(p, path, np.array([x[0], y[0], 1.0, 0.0]))
        want = np.array(c["coeffs"])
        V = np.vander(x, 6, increasing=True)
        assert ref.shape == (len(x), 2)
        assert np.max(np.abs(V @ coeffs - V @ want)) < 1e-9 * max(1.0, np.max(np.abs(y)))
        assert np.array_equal(coeffs == 0.0, want == 0.0)
        assert np.allclose(ref[:, 1], V @ coeffs, rtol=1e-12, atol=1e-12)
    # waypoint selection semantics (closest index, ≤ 20 ahead, short tail) equal the oracle's
    i = np.arange(200.0)
    path = np.stack([i, 0.5 * np.sin(0.05 * i)], 1)
    po = oracle.default_params()
    for ego in ([0, 0.1, 3, 0.02], [57.4, 0.3, 3, 0], [193.2, 0, 3, 0], [500.0, 0, 1, 0], [-20.0, 3.0, 1, 0]):
        c1, r1 = cilqr.local_plan(p, path, np.array(ego, dtype=float))
        c2, r2 = oracle.local_plan(po, path, np.array(ego, dtype=float))
        assert r1.shape == r2.shape and np.array_equal(r1[:, 0], r2[:, 0])
        V = np.vander(r1[:, 0], 6, increasing=True)
        assert np.max(np.abs(V @ c1 - V @ c2)) < 1e-9


def test_map_geom_matches_reference(cilqr):
    for c in load_golden("ref_gridmap.json")["geometry"]:
        g = cilqr.map_geom(*c["args"])
        assert (g.rows, g.cols, g.len_x, g.len_y) == (c["rows"], c["cols"], c["len_x"], c["len_y"])
    with pytest.raises(cilqr.CilqrError):
        cilqr.map_geom(0.0, 1.0, 0.1, 0, 0)


def test_argument_errors_do_not_need_a_device(cilqr):
    p = cilqr.default_params()
    p.num_states = 5  # BASELINE.json says nx=5; the reference model is nx=4 — refused, not guessed
    h = C.c_void_p()
    rc = cilqr.lib().cilqr_create(C.byref(p), 4, 50, 4, 0, C.byref(h))
    assert rc == -4 and b"num_states" in cilqr.lib().cilqr_last_error()
    p = cilqr.default_params()
    rc = cilqr.lib().cilqr_create(C.byref(p), 4, 1000, 4, 0, C.byref(h))
    assert rc == -1


@pytest.mark.skipif(_has_gpu(), reason="only meaningful where no GPU exists")
def test_no_cpu_fallback(cilqr):
    """Without a gfx950 device the product must fail loudly, never compute on the CPU."""
    with pytest.raises(cilqr.CilqrError) as e:
        cilqr.Solver(cilqr.default_params(), max_batch=4, max_horizon=50, max_obstacles=4)
    assert "error -2" in str(e.value)


def test_product_does_not_reference_oracle():
    """The product tree must not import, include or link anything under oracle/."""
    pkg = os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            hit = re.search(r"(from|import)\s+oracle|oracle\s*[/.]|liboracle|cilqr_oracle|_ref/", text)
            assert hit is None, (os.path.join(dirpath, f), hit.group(0))


def test_replay_tool_fails_loudly_without_a_device(cilqr, tmp_path):
    """bin/cilqr_replay on a box without a GPU: a message and exit code 1 — no CPU path, no crash."""
    import os
    import subprocess
    from conftest import PKG
    exe = os.path.join(PKG, "bin", "cilqr_replay")
    assert os.path.exists(exe), "bin/cilqr_replay not built"
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    log = tmp_path / "one_tick.log"
    log.write_text("cilqr-replay 1\nhorizon 10\npath 3\n0 0\n1 0\n2 0\ntick\nego 0 0 1 0\nobstacles 0\n")
    p = subprocess.run([exe, str(log)], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout == ""
    assert "device" in p.stderr.lower() or "hip" in p.stderr.lower()
