"""Seeded synthetic scenes for the BASELINE.json configs.  Host-side data generation only
(numpy); the RNG is numpy's PCG64 with seed 20250328 + config number, so every consumer (checker, HIP path, bench)
sees identical inputs.

Layouts are those of include/cilqr.h: x0 (B,4); U (B,2N); poly (B,6); xplan_fl (B,2); obs_pose (B,M,N,4) →
flattened (B,M,4N); obs_dim (B,M,N,2) → (B,M,2N); obs_weight (B,M) or None.
"""
import numpy as np

SEED0 = 20250328


def _default_local_plan():
    from . import local_plan
    return local_plan


def known_answer_scene(N, M, params, local_plan=None):
    """The scene of SURVEY.md §8(c): path (i, 0.5 sin(0.05 i)), x0 = (0, 0.1, 3, 0.02), static obstacles every 12 m."""
    local_plan = local_plan or _default_local_plan()
    i = np.arange(200.0)
    path = np.stack([i, 0.5 * np.sin(0.05 * i)], 1)
    x0 = np.array([[0.0, 0.1, 3.0, 0.02]])
    coeffs, ref = local_plan(params, path, x0[0])
    pose = np.zeros((1, M, N, 4))
    dim = np.zeros((1, M, N, 2))
    for o in range(M):
        pose[0, o, :, :] = [15 + 12 * o, (-1.0 if o % 2 else 0.8), 0.0, 0.1 * o]
        dim[0, o, :, :] = [4.79, 2.16]
    U = np.tile(np.array([[0.5, 0.0]]), (N, 1))
    U[N // 2:, 1] = 0.1
    return dict(N=N, M=M, x0=x0, U=U.reshape(1, 2 * N), poly=coeffs.reshape(1, 6),
                xplan_fl=np.array([[ref[0, 0], ref[-1, 0]]]),
                obs_pose=pose.reshape(1, M, 4 * N) if M else None, obs_dim=dim.reshape(1, M, 2 * N) if M else None,
                obs_weight=None)


def _paths_and_starts(rng, B, params, local_plan):
    A = rng.uniform(0.0, 1.5, B)
    om = rng.uniform(0.02, 0.08, B)
    ph = rng.uniform(0.0, 2 * np.pi, B)
    xs = np.arange(200.0)
    x0 = np.zeros((B, 4))
    y0 = A * np.sin(ph)
    slope0 = A * om * np.cos(ph)
    x0[:, 0] = 0.0
    x0[:, 1] = y0 + rng.uniform(-0.3, 0.3, B)
    x0[:, 2] = rng.uniform(1.0, 6.0, B)
    x0[:, 3] = np.arctan(slope0) + rng.uniform(-0.05, 0.05, B)
    poly = np.zeros((B, 6))
    xplan = np.zeros((B, 2))
    for b in range(B):
        path = np.stack([xs, A[b] * np.sin(om[b] * xs + ph[b])], 1)
        c, ref = local_plan(params, path, x0[b])
        poly[b] = c
        xplan[b] = (ref[0, 0], ref[-1, 0])
    return (A, om, ph), x0, poly, xplan


def _default_U(B, N):
    U = np.zeros((B, N, 2))
    U[:, :, 0] = 0.5
    U[:, N // 2:, 1] = 0.1
    return U.reshape(B, 2 * N)


def _obstacles(rng, curve, B, M, N, dt, speed_hi):
    A, om, ph = curve
    cx = rng.uniform(10.0, 45.0, (B, M))
    lat = rng.uniform(-2.5, 2.5, (B, M))
    tang = np.arctan(A[:, None] * om[:, None] * np.cos(om[:, None] * cx + ph[:, None]))
    cy = A[:, None] * np.sin(om[:, None] * cx + ph[:, None])
    head = tang + rng.uniform(-0.3, 0.3, (B, M))
    scale = rng.uniform(0.8, 1.2, (B, M))
    speed = rng.uniform(0.0, speed_hi, (B, M)) if speed_hi > 0 else np.zeros((B, M))
    # centre = point on the curve displaced along the curve normal
    px = cx - np.sin(tang) * lat
    py = cy + np.cos(tang) * lat
    t = np.arange(N) * dt
    pose = np.zeros((B, M, N, 4))
    pose[..., 0] = px[..., None] + (speed * np.cos(head))[..., None] * t
    pose[..., 1] = py[..., None] + (speed * np.sin(head))[..., None] * t
    pose[..., 2] = speed[..., None]
    pose[..., 3] = head[..., None]
    dim = np.zeros((B, M, N, 2))
    dim[..., 0] = (4.79 * scale)[..., None]
    dim[..., 1] = (2.16 * scale)[..., None]
    return pose, dim


def make_static(B, N, M, params, seed, local_plan=None):
    """C2 (N=50, M=4) / C5 (N=80, M=16) generator: sinusoidal path, M static obstacles."""
    local_plan = local_plan or _default_local_plan()
    rng = np.random.Generator(np.random.PCG64(seed))
    curve, x0, poly, xplan = _paths_and_starts(rng, B, params, local_plan)
    pose, dim = _obstacles(rng, curve, B, M, N, params.timestep, 0.0)
    return dict(N=N, M=M, x0=x0, U=_default_U(B, N), poly=poly, xplan_fl=xplan,
                obs_pose=pose.reshape(B, M, 4 * N) if M else None, obs_dim=dim.reshape(B, M, 2 * N) if M else None,
                obs_weight=None, curve=curve)


def make_c2(B=1024, params=None, local_plan=None):
    return make_static(B, 50, 4, params, SEED0 + 2, local_plan)


def make_c5(B=8192, params=None, local_plan=None, shard=0):
    return make_static(B, 80, 16, params, SEED0 + 5 + 1000 * shard, local_plan)


def make_c3(B=4096, params=None, local_plan=None, n_dyn=8, n_samples=32):
    """C3: n_dyn moving obstacles × n_samples Gaussian pose samples, each weighted 1/n_samples — the reference's own
    Obstacle path with M = n_dyn·n_samples objects and w_obstacle = 1/n_samples."""
    local_plan = local_plan or _default_local_plan()
    N = 50
    rng = np.random.Generator(np.random.PCG64(SEED0 + 3))
    curve, x0, poly, xplan = _paths_and_starts(rng, B, params, local_plan)
    pose, dim = _obstacles(rng, curve, B, n_dyn, N, params.timestep, 8.0)
    off = rng.normal(0.0, 1.0, (B, n_dyn, n_samples, 3)) * np.array([0.16, 0.16, 0.017])
    M = n_dyn * n_samples
    poses = np.repeat(pose[:, :, None, :, :], n_samples, axis=2)  # (B, n_dyn, S, N, 4)
    poses[..., 0] += off[..., 0][..., None]
    poses[..., 1] += off[..., 1][..., None]
    poses[..., 3] += off[..., 2][..., None]
    dims = np.repeat(dim[:, :, None, :, :], n_samples, axis=2)
    w = np.full((B, M), 1.0 / n_samples)
    return dict(N=N, M=M, x0=x0, U=_default_U(B, N), poly=poly, xplan_fl=xplan,
                obs_pose=poses.reshape(B, M, 4 * N), obs_dim=dims.reshape(B, M, 2 * N), obs_weight=w,
                # the same obstacles in compact form (cilqr_solve_batch_sampled): nominal trajectories + per-sample offsets
                nom_pose=pose.reshape(B, n_dyn, 4 * N), nom_dim=dim.reshape(B, n_dyn, 2 * N), offsets=off,
                sample_weight=1.0 / n_samples, curve=curve)


def materialise_samples(nom_pose, nom_dim, offsets, N):
    """The M = n_dyn·n_samples ordinary obstacles a compact sampled scene stands for (what the reference's own Obstacle path is
    given): (obs_pose (B, M, 4N), obs_dim (B, M, 2N), obs_weight (B, M))."""
    B, n_dyn, n_samples = offsets.shape[:3]
    pose = np.repeat(np.asarray(nom_pose).reshape(B, n_dyn, 1, N, 4), n_samples, axis=2).copy()
    pose[..., 0] += offsets[..., 0][..., None]
    pose[..., 1] += offsets[..., 1][..., None]
    pose[..., 3] += offsets[..., 2][..., None]
    dim = np.repeat(np.asarray(nom_dim).reshape(B, n_dyn, 1, N, 2), n_samples, axis=2)
    M = n_dyn * n_samples
    return pose.reshape(B, M, 4 * N), dim.reshape(B, M, 2 * N), np.full((B, M), 1.0 / n_samples)


POSE_SIGMA = np.array([0.16, 0.16, 0.017])  # ilqr/launch/Experiment.launch:9-11 (sigma_x, sigma_y, sigma_theta)


class TickSequence:
    """A closed-loop sequence of planner ticks over a batch of scenes — what a planner really hands the solver call after call,
    as opposed to one batch solved again and again.  Tick 0 is the benchmark batch itself (`kind` = "c3": make_c3; "static":
    make_static with the given seed).  `advance(X, U)` takes the results of the current tick and forms the next one the way the
    reference node does between two `run_step` calls:
      * ego <- X[:, 1], the state the accepted plan reaches after one timestep (the node re-reads odometry; this is its noise-free
        stand-in);
      * U: the result, kept UN-SHIFTED as the warm start (iLQR::control_seq persists across run_step calls, I/iLQR.cpp:253);
      * the local plan is re-fitted around the new ego (LocalPlanner pre-step on the same global path, I/LocalPlanner.cpp:25-117);
      * obstacles move on by one timestep along their headings (static ones stay);
      * "c3": every tick draws fresh pose noise for the samples with the launch file's sigmas (the node perturbs the
        obstacle poses anew in every callback, I/ilqr_uncertainty_node.cpp:82-110).
    Host-side numpy; the arrays of `inputs()` have the layouts of include/cilqr.h (compact sampled form for "c3")."""

    def __init__(self, kind, B, params, N=50, M=4, seed=None, local_plan=None, n_dyn=8, n_samples=32):
        self.kind, self.B, self.N, self.params = kind, B, N, params
        self.local_plan = local_plan or _default_local_plan()
        if kind == "c3":
            sc = make_c3(B, params, self.local_plan, n_dyn, n_samples)
            pose, dim = sc["nom_pose"].reshape(B, n_dyn, N, 4), sc["nom_dim"].reshape(B, n_dyn, N, 2)
            self.offsets, self.sample_weight, self.n_obs = sc["offsets"], sc["sample_weight"], n_dyn
            self.rng = np.random.Generator(np.random.PCG64(SEED0 + 3 + 7919))  # the noise of the later ticks
        else:
            sc = make_static(B, N, M, params, SEED0 + 2 if seed is None else seed, self.local_plan)
            pose, dim = sc["obs_pose"].reshape(B, M, N, 4), sc["obs_dim"].reshape(B, M, N, 2)
            self.offsets, self.sample_weight, self.n_obs = None, None, M
            self.rng = None
        self.curve = sc["curve"]
        self.x0, self.U, self.poly, self.xplan = sc["x0"].copy(), sc["U"].copy(), sc["poly"].copy(), sc["xplan_fl"].copy()
        self.p0 = pose[:, :, 0, :2].copy()                               # obstacle positions at the current tick
        self.speed, self.head = pose[:, :, 0, 2].copy(), pose[:, :, 0, 3].copy()
        self.dim = dim
        self.tick = 0

    def _poses(self):
        t = np.arange(self.N) * self.params.timestep
        pose = np.zeros((self.B, self.n_obs, self.N, 4))
        pose[..., 0] = self.p0[..., 0, None] + (self.speed * np.cos(self.head))[..., None] * t
        pose[..., 1] = self.p0[..., 1, None] + (self.speed * np.sin(self.head))[..., None] * t
        pose[..., 2] = self.speed[..., None]
        pose[..., 3] = self.head[..., None]
        return pose

    def inputs(self):
        B, N, n = self.B, self.N, self.n_obs
        d = dict(N=N, x0=self.x0, U=self.U, poly=self.poly, xplan_fl=self.xplan, tick=self.tick)
        pose = self._poses()
        if self.kind == "c3":
            d.update(M=n * self.offsets.shape[2], nom_pose=pose.reshape(B, n, 4 * N), nom_dim=self.dim.reshape(B, n, 2 * N),
                     offsets=self.offsets, sample_weight=self.sample_weight)
        else:
            d.update(M=n, obs_pose=pose.reshape(B, n, 4 * N), obs_dim=self.dim.reshape(B, n, 2 * N), obs_weight=None)
        return d

    def advance(self, X, U):
        """X (B, 4(N+1)), U (B, 2N): the solver's results for the current tick."""
        A, om, ph = self.curve
        dt = self.params.timestep
        self.x0 = np.ascontiguousarray(np.asarray(X).reshape(self.B, self.N + 1, 4)[:, 1, :])
        self.U = np.ascontiguousarray(np.asarray(U).reshape(self.B, 2 * self.N))
        self.p0[..., 0] += self.speed * np.cos(self.head) * dt
        self.p0[..., 1] += self.speed * np.sin(self.head) * dt
        xs = np.arange(200.0)
        for b in range(self.B):
            path = np.stack([xs, A[b] * np.sin(om[b] * xs + ph[b])], 1)
            c, ref = self.local_plan(self.params, path, self.x0[b])
            self.poly[b] = c
            self.xplan[b] = (ref[0, 0], ref[-1, 0])
        if self.kind == "c3":
            self.offsets = self.rng.normal(0.0, 1.0, self.offsets.shape) * POSE_SIGMA
        self.tick += 1


def make_c4(seed=SEED0 + 4, size=1024):
    """C4: source occupancy `size`×`size` float32 ({0,100} blobs + 2 % NaN) at 0.2 m; destination `size`×`size` at 0.1 m
    centred on the vehicle; 300-frame pose stream (theta 0→2π, position on a 20 m circle about the source centre)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    src = np.zeros((size, size), dtype=np.float32, order="F")
    n_blobs = 200
    ci = rng.integers(0, size, n_blobs)
    cj = rng.integers(0, size, n_blobs)
    rad = rng.integers(3, 24, n_blobs)
    ii, jj = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for k in range(n_blobs):
        src[(ii - ci[k]) ** 2 + (jj - cj[k]) ** 2 <= rad[k] ** 2] = 100.0
    src[rng.random((size, size)) < 0.02] = np.nan
    frames = 300
    th = np.linspace(0.0, 2 * np.pi, frames, endpoint=False)
    poses = np.stack([20.0 * np.cos(th), 20.0 * np.sin(th), th], 1)
    return dict(src=np.asfortranarray(src), src_geom=(size * 0.2, size * 0.2, 0.2, 0.0, 0.0),
                dst_geom=(size * 0.1, size * 0.1, 0.1, 0.0, 0.0), poses=poses)


def make_occupancy(rows, cols, seed, n_blobs=6, nan_frac=0.01):
    """A vehicle-frame occupancy layer (float32, 0 / 100 rectangles + a few unknown cells) for the uncertainty-cost tests and
    bench: what the map node's warp would hand to the blur."""
    rng = np.random.Generator(np.random.PCG64(seed))
    occ = np.zeros((rows, cols), dtype=np.float32)
    for _ in range(n_blobs):
        i, j = rng.integers(0, rows - 4), rng.integers(0, cols - 4)
        h, w = rng.integers(3, max(4, rows // 6)), rng.integers(3, max(4, cols // 6))
        occ[i:i + h, j:j + w] = 100.0
    occ[rng.random(occ.shape) < nan_frac] = np.nan
    return occ
