"""ctypes loader for the CPU oracle (oracle/build/liboracle.so) and the reference-built helpers
(oracle/_ref/libref_*.so).  TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg — never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_float_p = C.POINTER(C.c_float)


class Params(C.Structure):
    """Mirror of `cilqr_params` (include/cilqr.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("num_of_local_wpts", "poly_order", "horizon", "max_iterations", "num_states", "num_ctrls")] + \
               [(n, C.c_double) for n in
                ("desired_speed", "timestep", "tolerance", "w_acc", "w_yawrate", "w_pos", "w_vel", "w_obstacle",
                 "w_uncertainty", "q1_acc", "q2_acc", "q1_yawrate", "q2_yawrate", "q1_front", "q2_front", "q1_rear",
                 "q2_rear", "q1_uncertainty", "q2_uncertainty", "acc_max", "acc_min", "steer_angle_min",
                 "steer_angle_max", "wheelbase", "speed_max", "steer_control_max", "steer_control_min",
                 "throttle_control_max", "throttle_control_min", "t_safe", "s_safe_a", "s_safe_b", "ego_rad",
                 "ego_front", "ego_rear", "length", "width", "safe_length", "safe_width", "lamb_factor", "lamb_max")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MapGeom(C.Structure):
    """Mirror of `cilqr_map_geom` (include/cilqr.h)."""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("res", C.c_double), ("len_x", C.c_double),
                ("len_y", C.c_double), ("pos_x", C.c_double), ("pos_y", C.c_double)]


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def build(ref=True):
    """Compile the restatement (always) and, when /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref and os.path.isdir(REF_ROOT):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "build", "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.oracle_get_J.restype = C.c_double
        L.oracle_warp_costmap.restype = C.c_long
        L.oracle_blur.restype = C.c_long
        _lib = L
    return _lib


def ref_lib(name):
    """Returns the CDLL for oracle/_ref/libref_<name>.so or None when it was not built."""
    path = os.path.join(HERE, "_ref", "libref_%s.so" % name)
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    if name == "gridmap":
        L.ref_warp.restype = C.c_long
    return L


def default_params(horizon=None):
    p = Params()
    lib().oracle_params_default(C.byref(p))
    if horizon is not None:
        p.horizon = horizon
    return p


def default_control_seq(N):
    U = np.zeros(2 * N)
    lib().oracle_default_control_seq(N, _dp(U))
    return U


def solve(p, N, x0, U, coeffs, xf, xl, obs_pose=None, obs_dim=None, obs_weight=None):
    """One reference-order solve.  Returns dict(U, X, J, iters, status, trace)."""
    M = 0 if obs_pose is None else int(np.asarray(obs_pose).reshape(-1, 4 * N).shape[0])
    x0 = _f64(x0)
    U = _f64(U).copy()
    coeffs = _f64(coeffs)
    obs_pose = _f64(obs_pose)
    obs_dim = _f64(obs_dim)
    obs_weight = _f64(obs_weight)
    X = np.zeros(4 * (N + 1))
    J = C.c_double(0)
    st = C.c_int(0)
    trace = np.full(3 * p.max_iterations, np.nan)
    it = lib().oracle_solve(C.byref(p), N, M, _dp(x0), _dp(U), _dp(coeffs), C.c_double(xf), C.c_double(xl),
                            _dp(obs_pose), _dp(obs_dim), _dp(obs_weight), _dp(X), C.byref(J), C.byref(st), _dp(trace))
    return dict(U=U, X=X, J=J.value, iters=it, status=st.value, trace=trace.reshape(-1, 3)[:it])


def solve_batch(p, N, M, x0, U, poly, xplan_fl, obs_pose=None, obs_dim=None, obs_weight=None, threads=1):
    B = int(np.asarray(x0).reshape(-1, 4).shape[0])
    x0 = _f64(x0)
    U = _f64(U).copy()
    poly = _f64(poly)
    xplan_fl = _f64(xplan_fl)
    obs_pose = _f64(obs_pose)
    obs_dim = _f64(obs_dim)
    obs_weight = _f64(obs_weight)
    X = np.zeros((B, 4 * (N + 1)))
    J = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    lib().oracle_solve_batch(C.byref(p), B, N, M, _dp(x0), _dp(U), _dp(poly), _dp(xplan_fl), _dp(obs_pose),
                             _dp(obs_dim), _dp(obs_weight), _dp(X), _dp(J), iters.ctypes.data_as(c_int_p),
                             status.ctypes.data_as(c_int_p), int(threads))
    return dict(U=U.reshape(B, 2 * N), X=X, J=J, iters=iters, status=status)


def max_threads():
    """Threads the OpenMP loops should use: the CPUs this process may really run on — its affinity mask, cut down to the
    cgroup CPU quota when there is one (a GPU box hands a 1-GPU job a share of the host's cores; more threads than that are
    throttled in bursts and make timings noisy)."""
    n = int(lib().oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / float(f.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def local_plan(p, path, ego):
    path = _f64(path)
    ego = _f64(ego)
    P = path.size // 2
    coeffs = np.zeros(p.poly_order + 1)
    ref = np.zeros(2 * p.num_of_local_wpts)
    n = lib().oracle_local_plan(C.byref(p), _dp(path), P, _dp(ego), _dp(coeffs), _dp(ref))
    return coeffs, ref[:2 * n].reshape(n, 2)


def polyfit(x, y, degree):
    x = _f64(x)
    y = _f64(y)
    c = np.zeros(degree + 1)
    lib().oracle_polyfit(_dp(x), _dp(y), int(x.size), int(degree), _dp(c))
    return c


def map_geom(len_x, len_y, res, pos_x, pos_y):
    g = MapGeom()
    lib().oracle_map_geom_set(C.byref(g), C.c_double(len_x), C.c_double(len_y), C.c_double(res), C.c_double(pos_x),
                              C.c_double(pos_y))
    return g


def warp(src, sg, dg, vx, vy, vtheta, bbox=None, threads=1):
    """src: (rows, cols) float32 in Fortran (column-major) order.  Returns (dst F-ordered, n_out_of_range)."""
    src = np.asfortranarray(src, dtype=np.float32)
    assert src.shape == (sg.rows, sg.cols)
    dst = np.zeros((dg.rows, dg.cols), dtype=np.float32, order="F")
    bb = None
    if bbox is not None:
        bbox = np.asfortranarray(bbox, dtype=np.float32)
        bb = bbox.ctypes.data_as(c_float_p)
    n = lib().oracle_warp_costmap(src.ctypes.data_as(c_float_p), C.byref(sg), dst.ctypes.data_as(c_float_p),
                                  C.byref(dg), C.c_double(vx), C.c_double(vy), C.c_double(vtheta), bb, int(threads))
    return dst, int(n)


def blur(src, g, sin_t, cos_t, sigma_x, sigma_y, sigma_theta, index=0, threads=1):
    """src: (rows, cols) float32.  Returns (out F-ordered float32, counts (rows*cols,) int32 in linear cell order, n_empty)."""
    src = np.asfortranarray(src, dtype=np.float32)
    assert src.shape == (g.rows, g.cols)
    out = np.zeros((g.rows, g.cols), dtype=np.float32, order="F")
    cnt = np.zeros(g.rows * g.cols, dtype=np.int32)
    n = lib().oracle_blur(src.ctypes.data_as(c_float_p), C.byref(g), int(index), C.c_double(sin_t), C.c_double(cos_t),
                          C.c_double(sigma_x), C.c_double(sigma_y), C.c_double(sigma_theta), out.ctypes.data_as(c_float_p),
                          cnt.ctypes.data_as(c_int_p), int(threads))
    return out, cnt, int(n)


def occupancy_to_layer(occ):
    occ = np.ascontiguousarray(occ, dtype=np.int8).reshape(-1)
    out = np.zeros(occ.size, dtype=np.float32)
    lib().oracle_occupancy_to_layer(occ.ctypes.data_as(C.c_void_p), C.c_long(occ.size), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def layer_to_occupancy(layer, data_min, data_max):
    layer = np.ascontiguousarray(layer, dtype=np.float32).reshape(-1)
    out = np.zeros(layer.size, dtype=np.int8)
    lib().oracle_layer_to_occupancy(layer.ctypes.data_as(C.POINTER(C.c_float)), C.c_long(layer.size), C.c_float(data_min),
                                    C.c_float(data_max), out.ctypes.data_as(C.c_void_p))
    return out


class UncertaintyMap(C.Structure):
    """Mirror of `cilqr_uncertainty_map` (include/cilqr.h); HOST pointers on this side."""
    _fields_ = [("layer", C.c_void_p), ("geom", MapGeom), ("pose_x", C.c_double), ("pose_y", C.c_double),
                ("pose_theta", C.c_double), ("poses", C.c_void_p), ("layer_stride", C.c_int64), ("probes_l", C.c_int32),
                ("probes_w", C.c_int32)]


def uncertainty_map(layer, geom, pose=(0.0, 0.0, 0.0), probes=(3, 3), poses=None, batched=False):
    """layer: (rows, cols) float32, or (B, rows, cols) with batched=True; poses: None or (B, 3).  Returns (struct, keepalive)."""
    layer = np.asarray(layer, dtype=np.float32)
    if batched:
        flat = np.ascontiguousarray(np.stack([np.asfortranarray(a).flatten(order="F") for a in layer]))
        stride = flat.shape[1]
    else:
        flat = np.ascontiguousarray(np.asfortranarray(layer).flatten(order="F"))
        stride = 0
    m = UncertaintyMap()
    m.layer = flat.ctypes.data
    m.geom = geom
    m.pose_x, m.pose_y, m.pose_theta = pose
    keep = [flat]
    if poses is not None:
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        m.poses = poses.ctypes.data
        keep.append(poses)
    m.layer_stride = stride
    m.probes_l, m.probes_w = probes
    return m, keep


def layer_bilinear(layer, g, qx, qy):
    """oracle_layer_bilinear at arrays of positions → (value, d/dx, d/dy, ok)."""
    flat = np.ascontiguousarray(np.asfortranarray(layer, dtype=np.float32).flatten(order="F"))
    qx = np.atleast_1d(np.asarray(qx, dtype=np.float64))
    qy = np.atleast_1d(np.asarray(qy, dtype=np.float64))
    v, dx, dy = np.zeros(qx.size), np.zeros(qx.size), np.zeros(qx.size)
    ok = np.zeros(qx.size, dtype=bool)
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    for k in range(qx.size):
        ok[k] = bool(lib().oracle_layer_bilinear(flat.ctypes.data_as(c_float_p), C.byref(g), C.c_double(qx[k]), C.c_double(qy[k]),
                                                 C.byref(a), C.byref(b), C.byref(c)))
        v[k], dx[k], dy[k] = (a.value, b.value, c.value) if ok[k] else (np.nan, 0.0, 0.0)
    return v, dx, dy, ok


def uncertainty_cost(p, umap, states, b=0):
    """oracle_uncertainty_cost at (n, 4) states → (cost (n,), vx (n, 4), mx (n, 4, 4))."""
    states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 4)
    n = states.shape[0]
    cost, vx, mx = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 16))
    c = C.c_double(0)
    for k in range(n):
        lib().oracle_uncertainty_cost(C.byref(p), C.byref(umap), int(b), _dp(states[k]), C.byref(c), _dp(vx[k]), _dp(mx[k]))
        cost[k] = c.value
    return cost, vx, mx.reshape(n, 4, 4)


def solve_batch_unc(p, N, M, x0, U, poly, xplan_fl, obs_pose, obs_dim, obs_weight, umap, threads=1):
    B = int(np.asarray(x0).reshape(-1, 4).shape[0])
    x0 = _f64(x0)
    U = _f64(U).copy()
    poly = _f64(poly)
    xplan_fl = _f64(xplan_fl)
    obs_pose = _f64(obs_pose)
    obs_dim = _f64(obs_dim)
    obs_weight = _f64(obs_weight)
    X = np.zeros((B, 4 * (N + 1)))
    J = np.zeros(B)
    iters = np.zeros(B, dtype=np.int32)
    status = np.zeros(B, dtype=np.int32)
    lib().oracle_solve_batch_unc(C.byref(p), B, N, M, _dp(x0), _dp(U), _dp(poly), _dp(xplan_fl), _dp(obs_pose),
                                 _dp(obs_dim), _dp(obs_weight), C.byref(umap) if umap is not None else None, _dp(X), _dp(J),
                                 iters.ctypes.data_as(c_int_p), status.ctypes.data_as(c_int_p), int(threads))
    return dict(U=U.reshape(B, 2 * N), X=X, J=J, iters=iters, status=status)


def ref_linear(src, geom_args, qx, qy):
    """GridMap::atPosition(..., INTER_LINEAR) of the reference's own grid_map_core (oracle/_ref); None when not built."""
    L = ref_lib("gridmap")
    if L is None:
        return None
    src = np.asfortranarray(src, dtype=np.float32)
    qx = np.ascontiguousarray(qx, dtype=np.float64)
    qy = np.ascontiguousarray(qy, dtype=np.float64)
    out = np.zeros(qx.size, dtype=np.float32)
    ok = np.zeros(qx.size, dtype=np.int32)
    L.ref_linear(src.ctypes.data_as(c_float_p), *[C.c_double(v) for v in geom_args], int(qx.size), _dp(qx), _dp(qy),
                 out.ctypes.data_as(c_float_p), ok.ctypes.data_as(c_int_p))
    return out, ok.astype(bool)


def ref_blur(src, geom_args, sin_t, cos_t, sigma_x, sigma_y, sigma_theta, index=0):
    """The same through the reference's own grid_map_core + Eigen (oracle/_ref); None when _ref is not built."""
    L = ref_lib("gridmap")
    if L is None:
        return None
    g = map_geom(*geom_args)
    src = np.asfortranarray(src, dtype=np.float32)
    out = np.zeros((g.rows, g.cols), dtype=np.float32, order="F")
    cnt = np.zeros(g.rows * g.cols, dtype=np.int32)
    ell = np.zeros(3 * g.rows * g.cols)
    L.ref_blur(src.ctypes.data_as(c_float_p), *[C.c_double(v) for v in geom_args], int(index), C.c_double(sin_t), C.c_double(cos_t),
               C.c_double(sigma_x), C.c_double(sigma_y), C.c_double(sigma_theta), out.ctypes.data_as(c_float_p),
               cnt.ctypes.data_as(c_int_p), ell.ctypes.data_as(c_double_p))
    return out, cnt, ell.reshape(-1, 3)
