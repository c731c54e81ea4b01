"""cilqr_amd — thin Python host binding (ctypes) over the C-ABI of include/cilqr.h.

The product is `lib/libcilqr_hip.so` (hand-written HIP for gfx950 + the C-ABI); this module only loads it and
marshals numpy / torch buffers into plain pointers.  There is no CPU fallback here or in the library: if the
shared object is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("CILQR_LIB") or os.path.join(PKG_ROOT, "lib", "libcilqr_hip.so")  # CILQR_LIB: A/B builds when tuning

NX, NU, POLY = 4, 2, 6
FLAG_FAITHFUL_ITERS = 1
FLAG_GENERAL_ONLY = 2
EXIT_TOLERANCE, EXIT_LAMBDA_MAX, EXIT_MAX_ITER, EXIT_NUMERIC = 0, 1, 2, 3

# every symbol include/cilqr.h declares
ABI_SYMBOLS = (
    "cilqr_params_default", "cilqr_abi_version", "cilqr_device_count", "cilqr_last_error", "cilqr_default_control_seq",
    "cilqr_local_plan", "cilqr_local_plan_batch", "cilqr_local_plan_batch_device", "cilqr_create", "cilqr_destroy", "cilqr_host_alloc", "cilqr_host_free", "cilqr_solve_batch", "cilqr_solve_batch_device", "cilqr_solve_batch_sampled", "cilqr_solve_batch_sampled_device",
    "cilqr_argmin_device", "cilqr_wait", "cilqr_set_diag_buffer", "cilqr_set_pass_count_buffer", "cilqr_solve_family", "cilqr_solve_wavefronts", "cilqr_solve_sampled_wavefronts", "cilqr_debug_quu_inverse", "cilqr_debug_closest_sample", "cilqr_debug_blur_ellipse", "cilqr_warp_costmap", "cilqr_warp_costmap_device", "cilqr_warp_costmap_batch_device", "cilqr_blur_costmap", "cilqr_blur_costmap_device", "cilqr_map_geom_set",
    "cilqr_occupancy_to_layer", "cilqr_occupancy_to_layer_device", "cilqr_layer_to_occupancy", "cilqr_layer_to_occupancy_device",
    "cilqr_costmap_frame_device",
    "cilqr_set_uncertainty_map", "cilqr_set_uncertainty_map_device", "cilqr_clear_uncertainty_map", "cilqr_debug_uncertainty_cost",
    "cilqr_comm_unique_id", "cilqr_comm_init_rank", "cilqr_comm_destroy", "cilqr_comm_size", "cilqr_argmin_global_device", "cilqr_debug_select",
    "cilqr_create_multi", "cilqr_multi_destroy", "cilqr_multi_device_count", "cilqr_multi_handle", "cilqr_multi_solve_batch",
    "cilqr_shard_range", "cilqr_multi_uses_rccl", "cilqr_debug_fail_enqueue",
)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)


class Params(C.Structure):
    """`cilqr_params` — POD mirror of the reference's Parameters (I/Parameters.h:5-91)."""
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
    """`cilqr_map_geom` — geometry of a grid_map layer."""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("res", C.c_double), ("len_x", C.c_double),
                ("len_y", C.c_double), ("pos_x", C.c_double), ("pos_y", C.c_double)]


class UncertaintyMap(C.Structure):
    """`cilqr_uncertainty_map` — what the reference's (absent) Uncertainty object is constructed from."""
    _fields_ = [("layer", C.c_void_p), ("geom", MapGeom), ("pose_x", C.c_double), ("pose_y", C.c_double),
                ("pose_theta", C.c_double), ("poses", C.c_void_p), ("layer_stride", C.c_int64), ("probes_l", C.c_int32),
                ("probes_w", C.c_int32)]


class CilqrError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads lib/libcilqr_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CilqrError("%s not built — run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.cilqr_last_error.restype = C.c_char_p
        for name in ABI_SYMBOLS:
            getattr(L, name)  # AttributeError if a declared symbol is not exported
        _lib = L
    return _lib


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the C-ABI: 128 opaque bytes for `Solver.comm_init_rank` on every rank."""
    buf = (C.c_char * COMM_ID_BYTES)()
    _check(lib().cilqr_comm_unique_id(buf))
    return bytes(buf.raw)


def shard_range(B, n_shards, shard):
    """`cilqr_shard_range`: (first, count) of the contiguous, balanced shard a device / rank owns (host arithmetic only)."""
    first, count = C.c_int(), C.c_int()
    _check(lib().cilqr_shard_range(int(B), int(n_shards), int(shard), C.byref(first), C.byref(count)))
    return first.value, count.value


def pinned_empty(shape, dtype=np.float64):
    """A numpy array over page-locked memory from `cilqr_host_alloc` (never freed: keep and reuse it)."""
    L = lib()
    L.cilqr_host_alloc.restype = C.c_void_p
    L.cilqr_host_alloc.argtypes = [C.c_size_t]
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.cilqr_host_alloc(max(n, 1))
    if not p:
        raise CilqrError("cilqr_host_alloc failed: %s" % L.cilqr_last_error().decode())
    buf = (C.c_char * max(n, 1)).from_address(p)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pinned_copy(a):
    a = np.ascontiguousarray(a)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def _check(rc):
    if rc != 0:
        raise CilqrError("cilqr error %d: %s" % (rc, lib().cilqr_last_error().decode()))


def default_params(horizon=None):
    p = Params()
    lib().cilqr_params_default(C.byref(p))
    if horizon is not None:
        p.horizon = horizon
    return p


def default_control_seq(N):
    U = np.zeros(2 * N)
    _check(lib().cilqr_default_control_seq(int(N), U.ctypes.data_as(_dp)))
    return U


def local_plan(p, path, ego):
    """LocalPlanner pre-step.  path: (P, 2) waypoints.  Returns (coeffs[6], ref_traj (n, 2))."""
    path = np.ascontiguousarray(path, dtype=np.float64)
    ego = np.ascontiguousarray(ego, dtype=np.float64)
    coeffs = np.zeros(p.poly_order + 1)
    ref = np.zeros(2 * p.num_of_local_wpts)
    n = C.c_int(0)
    _check(lib().cilqr_local_plan(C.byref(p), path.ctypes.data_as(_dp), int(path.size // 2), ego.ctypes.data_as(_dp),
                                  coeffs.ctypes.data_as(_dp), ref.ctypes.data_as(_dp), C.byref(n)))
    return coeffs, ref[:2 * n.value].reshape(n.value, 2)


def map_geom(len_x, len_y, res, pos_x, pos_y):
    g = MapGeom()
    _check(lib().cilqr_map_geom_set(C.byref(g), C.c_double(len_x), C.c_double(len_y), C.c_double(res),
                                    C.c_double(pos_x), C.c_double(pos_y)))
    return g


def _np64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_dp):
    return None if a is None else a.ctypes.data_as(t)


def _vp(ptr):
    return C.c_void_p(int(ptr)) if ptr else None


class Solver:
    """One handle = one device = one host thread at a time (mirrors one reference `iLQR` object per batch slot)."""

    def __init__(self, params=None, max_batch=1024, max_horizon=50, max_obstacles=4, device=0):
        self.params = params if params is not None else default_params()
        self.max_batch, self.max_horizon, self.max_obstacles, self.device = max_batch, max_horizon, max_obstacles, device
        self._h = C.c_void_p()
        _check(lib().cilqr_create(C.byref(self.params), int(max_batch), int(max_horizon), int(max_obstacles),
                                  int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().cilqr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry point (synchronous) ----
    def solve_batch(self, N, x0, U, poly, xplan_fl, obs_pose=None, obs_dim=None, obs_weight=None, flags=0, out=None):
        """out: optional dict of preallocated arrays U (in/out: holds the warm start), X, J, iters, status — e.g. over pinned
        memory (`pinned_empty`); then U is used in place instead of being copied."""
        x0 = _np64(x0).reshape(-1, 4)
        B = x0.shape[0]
        U = _np64(U).reshape(B, 2 * N).copy() if out is None else out["U"]
        poly = _np64(poly).reshape(B, POLY)
        xplan_fl = _np64(xplan_fl).reshape(B, 2)
        M = 0
        if obs_pose is not None:
            obs_pose = _np64(obs_pose).reshape(B, -1, 4 * N)
            M = obs_pose.shape[1]
            obs_dim = _np64(obs_dim).reshape(B, M, 2 * N)
            if obs_weight is not None:
                obs_weight = _np64(obs_weight).reshape(B, M)
        if out is None:
            X, J = np.zeros((B, 4 * (N + 1))), np.zeros(B)
            iters, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        else:
            X, J, iters, status = out["X"], out["J"], out["iters"], out["status"]
        _check(lib().cilqr_solve_batch(self._h, B, int(N), int(M), _p(x0), _p(U), _p(poly), _p(xplan_fl), _p(obs_pose),
                                       _p(obs_dim), _p(obs_weight), _p(X), _p(J), _p(iters, _ip), _p(status, _ip),
                                       C.c_uint32(flags)))
        return dict(U=U, X=X, J=J, iters=iters, status=status)

    def solve_batch_sampled(self, N, x0, U, poly, xplan_fl, nom_pose, nom_dim, offsets, weight, flags=0):
        """Sampled obstacles in compact form: nom_pose (B, n_obs, 4N), nom_dim (B, n_obs, 2N), offsets (B, n_obs, S, 3)."""
        x0 = _np64(x0).reshape(-1, 4)
        B = x0.shape[0]
        U = _np64(U).reshape(B, 2 * N).copy()
        poly = _np64(poly).reshape(B, POLY)
        xplan_fl = _np64(xplan_fl).reshape(B, 2)
        offsets = _np64(offsets)
        n_obs, S = offsets.shape[1], offsets.shape[2]
        nom_pose = _np64(nom_pose).reshape(B, n_obs, 4 * N)
        nom_dim = _np64(nom_dim).reshape(B, n_obs, 2 * N)
        offsets = offsets.reshape(B, n_obs, S, 3)
        X = np.zeros((B, 4 * (N + 1)))
        J = np.zeros(B)
        iters = np.zeros(B, dtype=np.int32)
        status = np.zeros(B, dtype=np.int32)
        _check(lib().cilqr_solve_batch_sampled(self._h, B, int(N), int(n_obs), int(S), _p(x0), _p(U), _p(poly), _p(xplan_fl),
                                               _p(nom_pose), _p(nom_dim), _p(offsets), C.c_double(weight), _p(X), _p(J),
                                               _p(iters, _ip), _p(status, _ip), C.c_uint32(flags)))
        return dict(U=U, X=X, J=J, iters=iters, status=status)

    def solve_batch_sampled_device(self, stream, B, N, n_obs, n_samples, x0, U, poly, xplan_fl, nom_pose, nom_dim, offsets, weight,
                                   X_out, J_out, iters_out, status_out, flags=0):
        _check(lib().cilqr_solve_batch_sampled_device(self._h, _vp(stream), int(B), int(N), int(n_obs), int(n_samples), _vp(x0),
                                                      _vp(U), _vp(poly), _vp(xplan_fl), _vp(nom_pose), _vp(nom_dim), _vp(offsets),
                                                      C.c_double(weight), _vp(X_out), _vp(J_out), _vp(iters_out), _vp(status_out),
                                                      C.c_uint32(flags)))

    # ---- device-pointer entry point (asynchronous on `stream`) ----
    def solve_batch_device(self, stream, B, N, M, x0, U, poly, xplan_fl, obs_pose, obs_dim, obs_weight, X_out, J_out,
                           iters_out, status_out, flags=0):
        """All pointer arguments are integer device addresses (e.g. torch.Tensor.data_ptr()); 0/None = NULL."""
        _check(lib().cilqr_solve_batch_device(self._h, _vp(stream), int(B), int(N), int(M), _vp(x0), _vp(U), _vp(poly),
                                              _vp(xplan_fl), _vp(obs_pose), _vp(obs_dim), _vp(obs_weight), _vp(X_out),
                                              _vp(J_out), _vp(iters_out), _vp(status_out), C.c_uint32(flags)))

    # ---- batched LocalPlanner pre-step on the device ----
    def local_plan_batch(self, path, ego):
        """path: (P, 2) shared by all candidates, or (B, P, 2) one per candidate; ego: (B, 4).
        Returns dict(poly (B,6), xplan_fl (B,2), ref_traj (B, W, 2), n (B,))."""
        ego = _np64(ego).reshape(-1, 4)
        B = ego.shape[0]
        path = _np64(path)
        if path.ndim == 3:
            if path.shape[0] != B:
                raise CilqrError("local_plan_batch: path batch dimension does not match ego")
            P, stride = path.shape[1], 2 * path.shape[1]
        else:
            path = path.reshape(-1, 2)
            P, stride = path.shape[0], 0
        W = self.params.num_of_local_wpts
        poly = np.zeros((B, POLY))
        fl = np.zeros((B, 2))
        ref = np.zeros((B, W, 2))
        n = np.zeros(B, dtype=np.int32)
        _check(lib().cilqr_local_plan_batch(self._h, B, int(P), _p(path), C.c_int64(stride), _p(ego), _p(poly), _p(fl), _p(ref),
                                            _p(n, _ip)))
        return dict(poly=poly, xplan_fl=fl, ref_traj=ref, n=n)

    def local_plan_batch_device(self, stream, B, P, path, path_stride, ego, poly, xplan_fl, ref_traj=0, n_out=0):
        _check(lib().cilqr_local_plan_batch_device(self._h, _vp(stream), int(B), int(P), _vp(path), C.c_int64(path_stride), _vp(ego),
                                                   _vp(poly), _vp(xplan_fl), _vp(ref_traj), _vp(n_out)))

    def argmin_device(self, stream, B, J, out_pair):
        _check(lib().cilqr_argmin_device(self._h, _vp(stream), int(B), _vp(J), _vp(out_pair)))

    # ---- costmap-lookup uncertainty cost (iLQR::set_uncertainty_map / clear_uncertainty_map) ----
    def set_uncertainty_map(self, layer, geom, pose=(0.0, 0.0, 0.0), probes=(3, 3)):
        """layer: (rows, cols) float32 host array (the blurred occupancy); copied into a device buffer the handle owns."""
        flat = np.ascontiguousarray(np.asfortranarray(layer, dtype=np.float32).flatten(order="F"))
        assert flat.size == geom.rows * geom.cols
        m = UncertaintyMap()
        m.layer = flat.ctypes.data
        m.geom = geom
        m.pose_x, m.pose_y, m.pose_theta = pose
        m.probes_l, m.probes_w = probes
        _check(lib().cilqr_set_uncertainty_map(self._h, C.byref(m)))

    def set_uncertainty_map_device(self, layer_ptr, geom, pose=(0.0, 0.0, 0.0), probes=(3, 3), layer_stride=0, poses_ptr=0):
        """layer_ptr / poses_ptr: device addresses that stay valid while solves run (e.g. the blur kernel's output)."""
        m = UncertaintyMap()
        m.layer = int(layer_ptr)
        m.geom = geom
        m.pose_x, m.pose_y, m.pose_theta = pose
        m.poses = int(poses_ptr) if poses_ptr else None
        m.layer_stride = int(layer_stride)
        m.probes_l, m.probes_w = probes
        _check(lib().cilqr_set_uncertainty_map_device(self._h, C.byref(m)))

    def clear_uncertainty_map(self):
        _check(lib().cilqr_clear_uncertainty_map(self._h))

    def debug_uncertainty_cost(self, states):
        states = _np64(states).reshape(-1, 4)
        n = states.shape[0]
        cost, vx, mx = np.zeros(n), np.zeros((n, 2)), np.zeros((n, 3))
        _check(lib().cilqr_debug_uncertainty_cost(self._h, n, _p(states), _p(cost), _p(vx), _p(mx)))
        return cost, vx, mx

    # ---- cross-GPU exchange step (RCCL behind the C-ABI) ----
    def comm_init_rank(self, n_ranks, rank, id_bytes):
        """id_bytes: the 128 bytes of `comm_unique_id()` made on one rank and carried to the others by the host."""
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _check(lib().cilqr_comm_init_rank(self._h, int(n_ranks), int(rank), buf))

    def comm_size(self):
        return int(lib().cilqr_comm_size(self._h))

    def argmin_global_device(self, stream, B, J, index_offset, out_pair):
        _check(lib().cilqr_argmin_global_device(self._h, _vp(stream), int(B), _vp(J), C.c_int64(index_offset), _vp(out_pair)))

    def debug_select(self, triples):
        t = _np64(triples).reshape(-1, 3)
        out = np.zeros(2)
        _check(lib().cilqr_debug_select(self._h, int(t.shape[0]), _p(t), _p(out)))
        return float(out[0]), int(out[1])

    def set_diag_buffer(self, dev_ptr):
        _check(lib().cilqr_set_diag_buffer(self._h, _vp(dev_ptr)))

    def set_pass_count_buffer(self, dev_ptr):
        _check(lib().cilqr_set_pass_count_buffer(self._h, _vp(dev_ptr)))

    def solve_family(self, B, N, M):
        """`cilqr_solve_family`: lanes per solve for this shape — 64 = one wavefront per solve, less = the grouped family."""
        g = lib().cilqr_solve_family(self._h, int(B), int(N), int(M))
        if g < 0:
            _check(g)
        return g

    def solve_wavefronts(self, B, N, M):
        """`cilqr_solve_wavefronts`: wavefronts per solve of a static-obstacle batch on the one-wavefront family (1 or 2)."""
        w = lib().cilqr_solve_wavefronts(self._h, int(B), int(N), int(M))
        if w < 0:
            _check(w)
        return w

    def solve_sampled_wavefronts(self, B, N, n_obs):
        """`cilqr_solve_sampled_wavefronts`: wavefronts per solve sharing phase L of a sampled-obstacle solve (1, 2 or 4)."""
        w = lib().cilqr_solve_sampled_wavefronts(self._h, int(B), int(N), int(n_obs))
        if w < 0:
            _check(w)
        return w

    def debug_quu_inverse(self, Quu, lamb, general=True):
        Quu = _np64(Quu).reshape(-1, 4)
        lamb = _np64(lamb).reshape(-1)
        out = np.zeros_like(Quu)
        _check(lib().cilqr_debug_quu_inverse(self._h, int(Quu.shape[0]), _p(Quu), _p(lamb), _p(out), int(bool(general))))
        return out

    def debug_closest_sample(self, queries):
        """`cilqr_debug_closest_sample`: rows {poly[6], x_first, x_last, px, py} → int32 rows {search, full scan, by Newton}."""
        q = _np64(queries).reshape(-1, 10)
        out = np.zeros((q.shape[0], 3), dtype=np.int32)
        _check(lib().cilqr_debug_closest_sample(self._h, int(q.shape[0]), _p(q), out.ctypes.data_as(_ip)))
        return out

    def debug_blur_ellipse(self, abc):
        abc = _np64(abc).reshape(-1, 3)
        out = np.zeros_like(abc)
        _check(lib().cilqr_debug_blur_ellipse(self._h, int(abc.shape[0]), _p(abc), _p(out)))
        return out

    def wait(self):
        _check(lib().cilqr_wait(self._h))

    # ---- costmap warp ----
    def warp_costmap(self, src, src_geom, dst_geom, vx, vy, vtheta, bbox=None):
        """src: (rows, cols) float32 (any order; converted to column-major).  Returns (dst F-ordered, n_out_of_range)."""
        src = np.asfortranarray(src, dtype=np.float32)
        assert src.shape == (src_geom.rows, src_geom.cols)
        dst = np.zeros((dst_geom.rows, dst_geom.cols), dtype=np.float32, order="F")
        bb = None
        if bbox is not None:
            bbox = np.asfortranarray(bbox, dtype=np.float32)
            assert bbox.shape == dst.shape
            bb = bbox.ctypes.data_as(_fp)
        n = C.c_int64(0)
        _check(lib().cilqr_warp_costmap(self._h, src.ctypes.data_as(_fp), C.byref(src_geom), dst.ctypes.data_as(_fp),
                                        C.byref(dst_geom), C.c_double(vx), C.c_double(vy), C.c_double(vtheta), bb,
                                        C.byref(n)))
        return dst, int(n.value)

    def warp_costmap_device(self, stream, src, src_geom, dst, dst_geom, vx, vy, vtheta, bbox=0, n_oob=0):
        _check(lib().cilqr_warp_costmap_device(self._h, _vp(stream), _vp(src), C.byref(src_geom), _vp(dst),
                                               C.byref(dst_geom), C.c_double(vx), C.c_double(vy), C.c_double(vtheta),
                                               _vp(bbox), _vp(n_oob)))


    def warp_costmap_batch_device(self, stream, src, src_geom, dst, dst_geom, poses, bbox=0, n_oob=0):
        """poses: (K, 3) host array of (vx, vy, vtheta); dst: device address of K destination layers back to back."""
        poses = _np64(poses).reshape(-1, 3)
        _check(lib().cilqr_warp_costmap_batch_device(self._h, _vp(stream), _vp(src), C.byref(src_geom), _vp(dst), C.byref(dst_geom),
                                                     int(poses.shape[0]), _p(poses), _vp(bbox), _vp(n_oob)))

    def blur_costmap(self, src, geom, vtheta, sigma_x, sigma_y, sigma_theta, index=0):
        """src: (rows, cols) float32.  Returns (out F-ordered float32, counts (rows*cols,) int32)."""
        src = np.asfortranarray(src, dtype=np.float32)
        assert src.shape == (geom.rows, geom.cols)
        out = np.zeros((geom.rows, geom.cols), dtype=np.float32, order="F")
        cnt = np.zeros(geom.rows * geom.cols, dtype=np.int32)
        _check(lib().cilqr_blur_costmap(self._h, src.ctypes.data_as(_fp), C.byref(geom), int(index), C.c_double(vtheta),
                                        C.c_double(sigma_x), C.c_double(sigma_y), C.c_double(sigma_theta),
                                        out.ctypes.data_as(_fp), cnt.ctypes.data_as(_ip)))
        return out, cnt

    # ---- OccupancyGrid <-> layer ----
    def occupancy_to_layer(self, occ):
        occ = np.ascontiguousarray(occ, dtype=np.int8).reshape(-1)
        out = np.zeros(occ.size, dtype=np.float32)
        _check(lib().cilqr_occupancy_to_layer(self._h, occ.ctypes.data_as(C.c_void_p), C.c_int64(occ.size), _p(out, _fp)))
        return out

    def layer_to_occupancy(self, layer, data_min, data_max):
        layer = np.ascontiguousarray(layer, dtype=np.float32).reshape(-1)
        out = np.zeros(layer.size, dtype=np.int8)
        _check(lib().cilqr_layer_to_occupancy(self._h, _p(layer, _fp), C.c_int64(layer.size), C.c_float(data_min),
                                              C.c_float(data_max), out.ctypes.data_as(C.c_void_p)))
        return out

    def occupancy_to_layer_device(self, stream, occ, n_cells, layer):
        _check(lib().cilqr_occupancy_to_layer_device(self._h, _vp(stream), _vp(occ), C.c_int64(n_cells), _vp(layer)))

    def layer_to_occupancy_device(self, stream, layer, n_cells, data_min, data_max, occ):
        _check(lib().cilqr_layer_to_occupancy_device(self._h, _vp(stream), _vp(layer), C.c_int64(n_cells), C.c_float(data_min),
                                                     C.c_float(data_max), _vp(occ)))

    def costmap_frame_device(self, stream, global_layer, global_geom, vehicle_geom, vx, vy, vtheta, sigma_x, sigma_y,
                             sigma_theta, vehicle_layer, uncertainty_layer, occupancy_out=0, bbox=0, n_oob=0):
        _check(lib().cilqr_costmap_frame_device(self._h, _vp(stream), _vp(global_layer), C.byref(global_geom),
                                                C.byref(vehicle_geom), C.c_double(vx), C.c_double(vy), C.c_double(vtheta),
                                                _vp(bbox), C.c_double(sigma_x), C.c_double(sigma_y), C.c_double(sigma_theta),
                                                _vp(vehicle_layer), _vp(uncertainty_layer), _vp(occupancy_out), _vp(n_oob)))

    def blur_costmap_device(self, stream, src, geom, vtheta, sigma_x, sigma_y, sigma_theta, out, index=0, count_out=0):
        _check(lib().cilqr_blur_costmap_device(self._h, _vp(stream), _vp(src), C.byref(geom), int(index), C.c_double(vtheta),
                                               C.c_double(sigma_x), C.c_double(sigma_y), C.c_double(sigma_theta), _vp(out),
                                               _vp(count_out)))
