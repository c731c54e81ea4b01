#!/usr/bin/env python3
"""bench.py — CILQR solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: `cilqr_solve_batch_device` on BASELINE config 2 (B=1024 seeded synthetic
scenes, N=50, M=4 obstacles, fp64) with every input already resident in HBM, followed by the min-cost selection
(`cilqr_argmin_global_device`: with N > 1 ranks one RCCL all-gather of 24 bytes per rank — the (J, index) pair and the rank's
index offset — issued by the library on the handle's own communicator).  Each rank
owns its own shard of B scenes (weak scaling, no data-path collective).  The warm-start U is restored from a device copy
inside the timed region, because the solve overwrites it.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "uncertainty-aware-cilqr-for-trajectory-optimization_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Recorded (NOT live) figures: HBM bytes per launch of the dominant kernel and its issue counters come from rocprofv3 --pmc
# passes, which cannot be collected inside this process.  They are read from the newest profiles/rNN_pmc.json — written by
# tools/prof_summary.py from the committed rocprofv3 summaries — and every block that carries one is tagged with the file and
# the commit the profiled build was made from ("recorded_from", "recorded_head").  FETCH_SIZE is doubled as
# MI355X_MICROARCH.md §HBM prescribes for gfx950.


def _load_recorded():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    if not files:
        return {"workloads": {}, "recorded_head": None}, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


RECORDED, RECORDED_FROM = _load_recorded()
# the batch size each record was profiled at: a record describes that launch only
RECORDED_BATCH = {"c2": 1024, "c3": 4096, "c5": 8192, "warp": 1024, "warp16": 1024, "occ": 8192}


def recorded_traffic(workload, size):
    """(HBM bytes per launch, tag) of the dominant kernel from the recorded PMC passes, or (None, None)."""
    rec = RECORDED["workloads"].get(workload)
    if rec is None or RECORDED_BATCH.get(workload) != size or "hbm_bytes_per_launch" not in rec:
        return None, None
    return rec["hbm_bytes_per_launch"], {"recorded_from": RECORDED_FROM, "recorded_head": RECORDED.get("recorded_head"),
                                          "summary": rec.get("source"), "formula": "2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes"}


FP64_VALU_PEAK_TF = 78.6   # vector fp64 = half the 157.3 TF fp32 vector peak of MI355X_MICROARCH.md


_REAL_STDOUT = None


def emit(obj):
    """The one JSON line, on the process's original stdout."""
    line = json.dumps(obj) + "\n"
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line.encode())


def cpu_rate(call, units, min_seconds=2.0, reps=3):
    """Stable CPU-baseline timing: one untimed call (starts the OpenMP pool, faults the pages in), one calibration call, then
    `reps` repetitions each looping `call` for at least `min_seconds`; returns (median units/s, relative spread, calls per rep)."""
    call()
    t = time.perf_counter()
    call()
    one = max(time.perf_counter() - t, 1e-6)
    n = max(1, int(np.ceil(min_seconds / one)))
    rates = []
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n):
            call()
        rates.append(n * units / (time.perf_counter() - t))
    rates.sort()
    med = rates[len(rates) // 2]
    return med, (rates[-1] - rates[0]) / med, n


def algorithmic_bytes_per_solve(N, M):
    """bytes_in = 8(4 + 2N + 6 + 2 + 6MN), bytes_out = 8(2N + 4(N+1) + 1) + 8."""
    return 8 * (4 + 2 * N + 6 + 2 + 6 * M * N) + 8 * (2 * N + 4 * (N + 1) + 1) + 8


def bench_warp(args, rank, local_rank, world, dist, dev):
    """BASELINE config 4: 1024x1024 float32 occupancy map warped into a 1024x1024 vehicle-frame map, one frame per step,
    source/destination resident in HBM, pose stream of scenes.make_c4 (theta 0→2π)."""
    import cilqr_amd
    from cilqr_amd import scenes
    S = args.batch or 1024
    if S == 1024:
        c4 = scenes.make_c4()
    else:  # size sweep (not a BASELINE config): same geometry rules, random {0,100} payload with 2 % NaN
        rng = np.random.default_rng(4 + S)
        src_s = np.where(rng.random((S, S)) < 0.3, np.float32(100), np.float32(0))
        src_s[rng.random((S, S)) < 0.02] = np.nan
        th = np.linspace(0.0, 2 * np.pi, 300, endpoint=False)
        c4 = dict(src=np.asfortranarray(src_s.astype(np.float32)), src_geom=(S * 0.2, S * 0.2, 0.2, 0.0, 0.0),
                  dst_geom=(S * 0.1, S * 0.1, 0.1, 0.0, 0.0), poses=np.stack([20.0 * np.cos(th), 20.0 * np.sin(th), th], 1))
    sg, dg = cilqr_amd.map_geom(*c4["src_geom"]), cilqr_amd.map_geom(*c4["dst_geom"])
    solver = cilqr_amd.Solver(cilqr_amd.default_params(), max_batch=1, max_horizon=1, max_obstacles=0, device=local_rank)
    src = torch.from_numpy(np.ascontiguousarray(c4["src"].T)).to(dev)  # column-major payload
    K = max(1, args.frames)
    dst = torch.zeros(K * dg.rows * dg.cols, dtype=torch.float32, device=dev)
    oob = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    poses = c4["poses"]
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(k, rec=False):
        if rec:
            ev0[k].record()
        if K == 1:
            vx, vy, th = poses[(k + 7 * rank) % len(poses)]
            solver.warp_costmap_device(stream, src.data_ptr(), sg, dst.data_ptr(), dg, vx, vy, th, 0, 0)  # no out-of-range counter
        else:  # K consecutive frames of the pose stream in one launch
            idx = (np.arange(K) + K * k + 7 * rank) % len(poses)
            solver.warp_costmap_batch_device(stream, src.data_ptr(), sg, dst.data_ptr(), dg, poses[idx])
        if rec:
            ev1[k].record()
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if rank == 0:
        cells = dg.rows * dg.cols
        bytes_launch = 8 * cells * K  # 4 B read + 4 B write per destination cell
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = {"metric": "costmap warp frames/sec (%dx%d -> %dx%d)" % (sg.rows, sg.cols, dg.rows, dg.cols), "value": K * args.steps * world / elapsed, "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 payload / f64 index math",
               "data": "synthetic",
               "config": {"workload": ("BASELINE config 4: " if S == 1024 else "size sweep: ") +
                                      "%dx%d occupancy costmap warp, %d frame(s) per step and launch, maps resident in HBM" % (S, S, K)},
               "roofline": {"bound": "hbm", "kernel": "warp_batch_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": recorded_traffic("warp" if K == 1 else "warp%d" % K, S)[0],
                            "traffic_source": recorded_traffic("warp" if K == 1 else "warp%d" % K, S)[1],
                            "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch}}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            threads = O.max_threads()
            osg, odg = O.map_geom(*c4["src_geom"]), O.map_geom(*c4["dst_geom"])
            nf = 20
            res = {}

            def cpu_call():
                for k in range(nf):
                    res["w"], _ = O.warp(c4["src"], osg, odg, *poses[k], threads=threads)
            rate, spread, calls = cpu_rate(cpu_call, nf)
            want = res["w"]
            solver.warp_costmap_device(stream, src.data_ptr(), sg, dst.data_ptr(), dg, *poses[nf - 1], 0, oob.data_ptr())
            torch.cuda.synchronize()
            got = dst[:dg.rows * dg.cols].cpu().numpy().reshape(dg.cols, dg.rows).T
            out["cpu_baseline"] = {"value": rate, "unit": "frames/s", "cores": threads, "kind": "port", "spread": spread,
                                   "sample": "the first %d frames of the pose stream, OpenMP over cells; one untimed pass, then the "
                                             "median of 3 repetitions of %d passes (>= 2 s each)" % (nf, calls)}
            out["bit_exact_vs_oracle"] = bool(np.array_equal(got, want, equal_nan=True))
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_blur(args, rank, local_rank, world, dist, dev):
    """pose-uncertainty blur of the node's vehicle map.  Default size of the reference node (150 x 100 cells,
    M/src/local_costmap.cpp:132) with the launch-file sigmas, or --batch S for an S x S map at 0.1 m."""
    import cilqr_amd
    S = args.batch
    geom = (30.0, 20.0, 0.2, 15.0, 0.0) if not S else (S * 0.1, S * 0.1, 0.1, 5.0, -3.0)
    sig = (0.16, 0.16, 0.017)
    g = cilqr_amd.map_geom(*geom)
    rng = np.random.default_rng(41 + rank)
    src_h = rng.integers(0, 101, (g.rows, g.cols)).astype(np.float32)
    solver = cilqr_amd.Solver(cilqr_amd.default_params(), max_batch=1, max_horizon=1, max_obstacles=0, device=local_rank)
    src = torch.from_numpy(np.ascontiguousarray(src_h.T)).to(dev)
    out = torch.zeros(g.rows * g.cols, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(k, rec=False):
        if rec:
            ev0[k].record()
        solver.blur_costmap_device(stream, src.data_ptr(), g, 0.3 + 0.01 * k, *sig, out.data_ptr())
        if rec:
            ev1[k].record()
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if rank == 0:
        cells = g.rows * g.cols
        bytes_launch = 8 * cells
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        outj = {"metric": "uncertainty-blur frames/sec (%dx%d cells)" % (g.rows, g.cols), "value": args.steps * world / elapsed,
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64 (float eigen-solve), f32 payload", "data": "synthetic",
                "config": {"workload": "SURVEY 8f-1: thrust_propagateUncertainty replacement, %dx%d cells, sigmas %s" % (g.rows, g.cols, sig)},
                "roofline": {"bound": "hbm", "kernel": "blur_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                             "algorithmic_bytes_per_launch": bytes_launch,
                             "note": "fp64-VALU-bound: ~20-400 tested cells and ~20-80 exp() per output cell"}}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            threads = O.max_threads()
            og = O.map_geom(*geom)
            t1 = time.perf_counter()
            want, wcnt, _ = O.blur(src_h, og, np.sin(0.3), np.cos(0.3), *sig, threads=threads)
            cpu_s = time.perf_counter() - t1
            solver.blur_costmap_device(stream, src.data_ptr(), g, 0.3, *sig, out.data_ptr())
            torch.cuda.synchronize()
            got = out.cpu().numpy().reshape(g.cols, g.rows).T
            outj["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": threads, "kind": "port", "sample": "one frame, OpenMP over cells"}
            outj["max_abs_diff_vs_oracle"] = float(np.nanmax(np.abs(got - want)))
        emit(outj)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def _timed(step, args, dist, dev):
    """Warm-up, then args.steps timed steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, mean of the
    HIP-event durations the step recorded around its dominant kernel, in ms)."""
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    for k in range(args.warmup):
        step(k, None, None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, ev0[k], ev1[k])
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))


def _line(metric, unit, value, args, world, elapsed, dtype, workload, roofline):
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic", "config": {"workload": workload}, "roofline": roofline}


def bench_occ(args, rank, local_rank, world, dist, dev):
    """OccupancyGrid -> layer -> OccupancyGrid on an S x S grid (default 8192: 67 M cells, 5 B per cell and
    direction), both buffers resident in HBM.  A step = both conversions; the roofline figure is the layer->occupancy one."""
    import cilqr_amd
    S = args.batch or 8192
    n = S * S
    solver = cilqr_amd.Solver(cilqr_amd.default_params(), max_batch=1, max_horizon=1, max_obstacles=0, device=local_rank)
    rng = np.random.default_rng(51 + rank)
    occ_h = rng.integers(-1, 101, n, dtype=np.int8)
    occ = torch.from_numpy(occ_h).to(dev)
    layer = torch.zeros(n, dtype=torch.float32, device=dev)
    back = torch.zeros(n, dtype=torch.int8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    first = []

    def step(k, e0, e1):
        if e0 is not None:
            first.append((torch.cuda.Event(enable_timing=True), e0))
            first[-1][0].record()
        solver.occupancy_to_layer_device(stream, occ.data_ptr(), n, layer.data_ptr())
        if e0 is not None:
            e0.record()
        solver.layer_to_occupancy_device(stream, layer.data_ptr(), n, -1.0, 100.0, back.data_ptr())
        if e1 is not None:
            e1.record()
    elapsed, kern_ms = _timed(step, args, dist, dev)
    if rank == 0:
        bytes_launch = 5 * n
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = _line("OccupancyGrid<->layer cells/sec (%dx%d, both directions per step)" % (S, S), "cells/s",
                    n * args.steps * world / elapsed, args, world, elapsed, "i8 / f32",
                    "SURVEY 8f-4: fromOccupancyGrid + toOccupancyGrid(-1, 100) round trip, %d cells, buffers resident in HBM" % n,
                    {"bound": "hbm", "kernel": "layer_to_occ_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": recorded_traffic("occ", S)[0], "traffic_source": recorded_traffic("occ", S)[1],
                     "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch})
        to_layer_ms = float(np.mean([a.elapsed_time(b) for a, b in first]))
        out["occ_to_layer_kernel"] = {"kernel_ms": to_layer_ms, "achieved": bytes_launch / (to_layer_ms * 1e-3) / 1e9, "unit": "GB/s",
                                      "frac": bytes_launch / (to_layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out["round_trip_exact"] = bool(torch.equal(back, occ))
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            ns = min(n, 1 << 26)
            t1 = time.perf_counter()
            lay = O.occupancy_to_layer(occ_h[:ns])
            O.layer_to_occupancy(lay, -1.0, 100.0)
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "cells/s", "cores": 1, "kind": "port",
                                   "sample": "%d cells, both directions, scalar loops as in the reference" % ns}
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_frame(args, rank, local_rank, world, dist, dev):
    """One frame of the map node's odometry callback per step (cilqr_costmap_frame_device: warp + bbox override -> blur ->
    OccupancyGrid out).  Default: the node's own sizes (1506x1506 global map at 0.2 m, 150x100 vehicle map,
    M/src/local_costmap.cpp:119,132); --batch S: config-4 shape, S x S vehicle map at 0.1 m out of a 1024x1024 source."""
    import cilqr_amd
    from cilqr_amd import scenes
    S = args.batch
    rng = np.random.default_rng(61 + rank)
    if S:
        c4 = scenes.make_c4()
        sgeo, dgeo = c4["src_geom"], (S * 0.1, S * 0.1, 0.1, 0.0, 0.0)
        src_h = c4["src"]
        poses = [(vx, vy, th) for vx, vy, th in c4["poses"]]
        shrink = S * 0.1 / c4["dst_geom"][0]
        assert shrink <= 1.0, "--batch larger than config 4's 1024"
    else:
        sgeo, dgeo = (301.2, 301.2, 0.2, 0.0, 0.0), (30.0, 20.0, 0.2, 10.0 - 5, 0.0)
        src_h = np.zeros((1506, 1506), dtype=np.float32)
        for _ in range(400):
            i, j = rng.integers(0, 1450, 2)
            src_h[i:i + rng.integers(3, 50), j:j + rng.integers(3, 50)] = 100.0
        src_h[rng.random(src_h.shape) < 0.02] = np.nan
        poses = [(60 * np.cos(a), 60 * np.sin(a), a + np.pi / 2) for a in np.linspace(0, 2 * np.pi, 300, endpoint=False)]
    sg, dg = cilqr_amd.map_geom(*sgeo), cilqr_amd.map_geom(*dgeo)
    solver = cilqr_amd.Solver(cilqr_amd.default_params(), max_batch=1, max_horizon=1, max_obstacles=0, device=local_rank)
    src = torch.from_numpy(np.ascontiguousarray(src_h.T)).to(dev)
    nd = dg.rows * dg.cols
    veh = torch.zeros(nd, dtype=torch.float32, device=dev)
    unc = torch.zeros(nd, dtype=torch.float32, device=dev)
    occ = torch.zeros(nd, dtype=torch.int8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sig = (0.16, 0.16, 0.017)

    def step(k, e0, e1):
        vx, vy, th = poses[(k + 7 * rank) % len(poses)]
        if e0 is not None:
            e0.record()
        solver.costmap_frame_device(stream, src.data_ptr(), sg, dg, vx, vy, th, *sig, veh.data_ptr(), unc.data_ptr(), occ.data_ptr())
        if e1 is not None:
            e1.record()
    elapsed, kern_ms = _timed(step, args, dist, dev)
    if rank == 0:
        bytes_launch = 13 * nd  # warp: 4 read + 4 write; blur: 4 read (+ neighbours from cache) + 4 + 1 write
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = _line("costmap frames/sec (warp + blur + OccupancyGrid, %dx%d vehicle map)" % (dg.rows, dg.cols), "frames/s",
                    args.steps * world / elapsed, args, world, elapsed, "f64 index math and blur weights, f32 layers, i8 grid",
                    "one odomCallback frame (M/src/local_costmap.cpp:172-305): %dx%d source -> %dx%d vehicle map, sigmas %s, maps "
                    "resident in HBM" % (sg.rows, sg.cols, dg.rows, dg.cols, sig),
                    {"bound": "hbm", "kernel": "warp_kernel + blur_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch,
                     "note": "both launches of one frame inside the event pair; the blur is fp64-VALU-bound, see DESIGN.md §4.5"})
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            threads = O.max_threads()
            osg, odg = O.map_geom(*sgeo), O.map_geom(*dgeo)
            vx, vy, th = poses[0]
            t1 = time.perf_counter()
            w, _ = O.warp(src_h, osg, odg, vx, vy, th, threads=threads)
            b, _, _ = O.blur(w, odg, np.sin(th), np.cos(th), *sig, threads=threads)
            o = O.layer_to_occupancy(b.reshape(-1, order="F"), 0.0, 100.0)
            cpu_s = time.perf_counter() - t1
            solver.costmap_frame_device(stream, src.data_ptr(), sg, dg, vx, vy, th, *sig, veh.data_ptr(), unc.data_ptr(), occ.data_ptr())
            torch.cuda.synchronize()
            out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "one frame, OpenMP over cells (warp, blur), scalar conversion"}
            out["occupancy_cells_equal_oracle"] = float(np.mean(occ.cpu().numpy() == o))
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_plan(args, rank, local_rank, world, dist, dev):
    """the LocalPlanner pre-step for B candidate ego poses per step (default 1024) on one shared 200-waypoint
    global path, everything resident in HBM."""
    import cilqr_amd
    B, P = args.batch or 1024, 200
    p = cilqr_amd.default_params()
    solver = cilqr_amd.Solver(p, max_batch=1, max_horizon=1, max_obstacles=0, device=local_rank)
    rng = np.random.default_rng(71 + rank)
    x = 37.25 + np.cumsum(rng.uniform(0.8, 1.2, P))
    path_h = np.stack([x, 1.2 * np.sin(0.05 * x)], axis=1)
    k = rng.integers(0, P, B)
    ego_h = np.stack([x[k] + rng.uniform(-0.4, 0.4, B), path_h[k, 1] + rng.uniform(-1, 1, B), rng.uniform(0, 8, B),
                      rng.uniform(-1, 1, B)], axis=1)
    path, ego = torch.from_numpy(path_h).to(dev), torch.from_numpy(ego_h).to(dev)
    poly = torch.zeros(B, 6, dtype=torch.float64, device=dev)
    fl = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(k, e0, e1):
        if e0 is not None:
            e0.record()
        solver.local_plan_batch_device(stream, B, P, path.data_ptr(), 0, ego.data_ptr(), poly.data_ptr(), fl.data_ptr())
        if e1 is not None:
            e1.record()
    elapsed, kern_ms = _timed(step, args, dist, dev)
    if rank == 0:
        bytes_launch = B * (32 + 64) + 16 * P  # ego in, poly + first/last out, the shared path once
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        out = _line("LocalPlanner fits/sec (B candidates, 20x6 column-pivoted QR each)", "fits/s", B * args.steps * world / elapsed,
                    args, world, elapsed, "f64", "SURVEY 8f-2: %d candidate ego poses per step on one %d-waypoint path" % (B, P),
                    {"bound": "hbm", "kernel": "local_plan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch,
                     "note": "serial-chain bound: one lane per fit, sums kept in the host pre-step's order"})
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            po = O.default_params(50)
            ns = min(B, 4096)
            t1 = time.perf_counter()
            want = np.array([O.local_plan(po, path_h, ego_h[b])[0] for b in range(ns)])
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "fits/s", "cores": 1, "kind": "port",
                                   "sample": "the first %d candidates, one thread, ctypes call per fit" % ns}
            out["coefficients_bit_equal_fraction"] = float(np.mean(np.all(poly.cpu().numpy()[:ns] == want, axis=1)))
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_c1(args, rank, local_rank, world, dist, dev):
    """BASELINE config 1 — the drop-in case: ONE solve per call (B=1, N=30, M=2; the known-answer scene of SURVEY §8c) through
    the host-buffer entry point `cilqr_solve_batch`, i.e. what replaces the reference's timed `run_step`
    (I/ilqr_uncertainty_node.cpp:117-122): H2D of the inputs + the solve + D2H of U, X, J every call.  A step = one call;
    the latency is the median over --steps calls (default here: 1000)."""
    import cilqr_amd
    from cilqr_amd import scenes
    N, M = 30, 2
    if args.batch:  # other single-solve shapes: --batch encodes the horizon (50 → M=4, 80 → M=16: the other survey scenes)
        N = args.batch
        M = {30: 2, 50: 4, 80: 16}.get(N, 4)
    steps = args.steps if args.steps != 50 else 1000
    p = cilqr_amd.default_params(N)
    sc = scenes.known_answer_scene(N, M, p)
    solver = cilqr_amd.Solver(p, max_batch=1, max_horizon=N, max_obstacles=M, device=local_rank)
    lat = []
    for k in range(args.warmup + steps):
        t = time.perf_counter()
        got = solver.solve_batch(N, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"])
        if k >= args.warmup:
            lat.append(time.perf_counter() - t)
    lat = np.array(lat)
    # device-side part of the same call: the kernel pair between two events on a torch stream, inputs resident
    dvt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x0, U0, poly, xpl, pose, dim = (dvt(sc[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim"))
    U = U0.clone()
    X = torch.zeros(4 * (N + 1), dtype=torch.float64, device=dev)
    J = torch.zeros(1, dtype=torch.float64, device=dev)
    it = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(k, e0, e1):
        U.copy_(U0)
        if e0 is not None:
            e0.record()
        solver.solve_batch_device(stream, 1, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(),
                                  dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), it.data_ptr(), st.data_ptr())
        if e1 is not None:
            e1.record()
    a2 = argparse.Namespace(steps=min(steps, 200), warmup=args.warmup)
    elapsed, kern_ms = _timed(step, a2, dist, dev)
    if rank == 0:
        med = float(np.median(lat))
        bytes_launch = algorithmic_bytes_per_solve(N, M)
        out = {"metric": "single CILQR solve latency through cilqr_solve_batch (B=1, N=%d, M=%d)" % (N, M), "value": 1e3 * med,
               "unit": "ms", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * med,
               "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE config 1 shape on the GPU: one solve per call, host buffers in and out (pageable), "
                                      "known-answer scene of SURVEY 8(c)", "horizon": N, "obstacles": M,
                          "iterations": int(got["iters"][0]), "exit": int(got["status"][0])},
               "latency_ms": {"median": 1e3 * med, "p10": 1e3 * float(np.percentile(lat, 10)), "p90": 1e3 * float(np.percentile(lat, 90)),
                              "min": 1e3 * float(lat.min())},
               "roofline": {"bound": "hbm", "kernel": "cilqr_solve_share_kernel<%d>" % solver.solve_wavefronts(1, N, M) if solver.solve_wavefronts(1, N, M) >= 2 else "cilqr_solve_kernel",
                            "achieved": bytes_launch / (kern_ms * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_launch / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch,
                            "note": "one solve on one CU (two wavefronts while it linearises, one otherwise): pure serial-chain latency, the rest of the chip idle"}}
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as O
            O.build(ref=False)
            po = O.default_params(N)
            res = {}

            def cpu_call():
                res["w"] = O.solve_batch(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, threads=1)
            rate, spread, calls = cpu_rate(cpu_call, 1)
            out["cpu_baseline"] = {"value": 1e3 / rate, "unit": "ms", "cores": 1, "kind": "port",
                                   "sample": "the same scene, one thread; median of 3 repetitions of %d solves" % calls, "spread": spread}
            out["max_abs_du_vs_oracle"] = float(np.max(np.abs(got["U"] - res["w"]["U"])))
            out["iters_equal_oracle"] = bool(int(got["iters"][0]) == int(res["w"]["iters"][0]))
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="solves per GPU per step (default: the config's own size)")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5", "warp", "blur", "occ", "frame", "plan"],
                    help="c2 (default, the config BASELINE.json's metric is quoted on): B=1024 N=50 M=4; c3: B=4096 N=50, 8x32 "
                         "sampled obstacles; c5: B=8192 per GPU N=80 M=16; warp: config 4, 1024x1024 costmap frames")
    ap.add_argument("--frames", type=int, default=1, help="warp only: K frames per launch (cilqr_warp_costmap_batch_device)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ticks", action="store_true",
                    help="solver workloads: every step solves the NEXT tick of a closed-loop planner sequence (ego moved one step along the "
                         "accepted plan, warm-started controls, re-fitted local plan, obstacles moved on, fresh pose noise) instead of the "
                         "same batch again; the sequence is generated before the timed region and lies in HBM.  Default for --workload c3.")
    ap.add_argument("--repeat-batch", action="store_true", help="c3: the old behaviour, one batch solved again and again")
    ap.add_argument("--materialised", action="store_true",
                    help="c3 only: pass the 256 sampled obstacles as 256 ordinary obstacle tables (cilqr_solve_batch_device) instead "
                         "of the compact nominal + offsets form (cilqr_solve_batch_sampled_device)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line (rank 0).  Libraries underneath write there too (RCCL prints a version banner when a
    # communicator is first made): everything but that line goes to stderr, at the file-descriptor level.
    sys.stdout.flush()
    global _REAL_STDOUT
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank): RCCL process group
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import cilqr_amd
    from cilqr_amd import scenes
    from cilqr_amd.dist import init_comm, select_min_cost_device

    dev = torch.device("cuda", local_rank)
    if args.workload == "warp":
        return bench_warp(args, rank, local_rank, world, dist, dev)
    if args.workload == "blur":
        return bench_blur(args, rank, local_rank, world, dist, dev)
    if args.workload in ("occ", "frame", "plan", "c1"):
        return {"occ": bench_occ, "frame": bench_frame, "plan": bench_plan, "c1": bench_c1}[args.workload](args, rank, local_rank, world, dist, dev)
    if args.workload == "c2":
        B, N, M = args.batch or 1024, 50, 4
        p = cilqr_amd.default_params(N)
        sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2 + 1000 * rank)  # rank 0 == scenes.make_c2(B)
        wl = "BASELINE config 2: B=%d CILQR solves per GPU per step, N=50, M=4 static obstacles" % B
    elif args.workload == "c5":
        B, N, M = args.batch or 8192, 80, 16
        p = cilqr_amd.default_params(N)
        sc = scenes.make_c5(B, p, shard=rank)
        wl = "BASELINE config 5 shard: B=%d CILQR solves per GPU per step, N=80, M=16 static obstacles" % B
    else:
        B, N = args.batch or 4096, 50
        p = cilqr_amd.default_params(N)
        sc = scenes.make_c3(B, p)
        M = sc["M"]
        wl = ("BASELINE config 3: B=%d CILQR solves per GPU per step, N=50, 8 moving obstacles x 32 Gaussian samples (M=256, weight 1/32), "
              % B) + ("materialised obstacle tables" if args.materialised else "compact form (nominal trajectories + sample offsets)")
    solver = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=local_rank)
    if dist is not None:  # one RCCL communicator over the ranks, owned by the handle: the exchange step is a C-ABI call
        init_comm(solver, dist)

    def dv(a, dtype=torch.float64):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(dev)
    x0, U0, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
    sampled = args.workload == "c3" and not args.materialised
    if sampled:  # compact form: nominal trajectories + per-sample offsets (cilqr_solve_batch_sampled_device)
        pose, dim, offs = dv(sc["nom_pose"]), dv(sc["nom_dim"]), dv(sc["offsets"])
        n_dyn, n_smp = sc["offsets"].shape[1], sc["offsets"].shape[2]
        wts = None
    else:
        pose, dim = dv(sc["obs_pose"]), dv(sc["obs_dim"])
        wts = dv(sc["obs_weight"]) if sc["obs_weight"] is not None else None
    U = U0.clone()
    use_ticks = (args.ticks or args.workload == "c3") and not args.repeat_batch and not args.materialised
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev)
    J = torch.zeros(B, dtype=torch.float64, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    pair = torch.zeros(2, dtype=torch.float64, device=dev)
    passes = torch.zeros(B, dtype=torch.int32, device=dev)
    solver.set_pass_count_buffer(passes.data_ptr())  # executed backward+forward passes per solve (one int32 store per solve)
    stream = torch.cuda.current_stream().cuda_stream
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    tick_in = []  # --ticks: the device-resident inputs of tick 0, 1, … (warm-up ticks first)
    last_host = None  # … and the host arrays of the last one (the oracle's inputs for the parity figure)
    cur = {"x0": x0, "U0": U0, "poly": poly, "xpl": xpl, "pose": pose, "dim": dim, "offs": offs if sampled else None}

    def launch(slv):
        x0, poly, xpl, pose, dim, offs = cur["x0"], cur["poly"], cur["xpl"], cur["pose"], cur["dim"], cur["offs"]
        if sampled:
            slv.solve_batch_sampled_device(stream, B, N, n_dyn, n_smp, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                           pose.data_ptr(), dim.data_ptr(), offs.data_ptr(), sc["sample_weight"], X.data_ptr(),
                                           J.data_ptr(), iters.data_ptr(), status.data_ptr())
        else:
            slv.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(),
                                   dim.data_ptr(), wts.data_ptr() if wts is not None else 0, X.data_ptr(), J.data_ptr(),
                                   iters.data_ptr(), status.data_ptr())

    tick_no = [0]

    def step(k=None):
        if use_ticks:  # the next tick of the sequence: another batch every step
            cur.update(tick_in[tick_no[0]])
            tick_no[0] += 1
        U.copy_(cur["U0"])
        if k is not None:
            ev0[k].record()
        launch(solver)
        if k is not None:
            ev1[k].record()
        # local argmin + ONE ncclAllGather of 24 bytes per rank + the pick, all enqueued by cilqr_argmin_global_device
        return select_min_cost_device(solver, stream, B, J.data_ptr(), rank * B, pair)

    if use_ticks:
        # Closed-loop generation, untimed: tick t + 1 comes from the results of tick t (scenes.TickSequence), so the whole sequence
        # of warmup + steps ticks is solved once here and its inputs are kept on the device for the timed replay.
        ts = scenes.TickSequence("c3" if sampled else "static", B, p, N=N, M=M,
                                 seed={"c2": scenes.SEED0 + 2 + 1000 * rank, "c5": scenes.SEED0 + 5 + 1000 * rank}.get(args.workload))
        for t in range(args.warmup + args.steps):
            i = ts.inputs()
            d = {"x0": dv(i["x0"]), "U0": dv(i["U"]), "poly": dv(i["poly"]), "xpl": dv(i["xplan_fl"])}
            if sampled:
                d.update(pose=dv(i["nom_pose"]), dim=dv(i["nom_dim"]), offs=dv(i["offsets"]))
            else:
                d.update(pose=dv(i["obs_pose"]), dim=dv(i["obs_dim"]), offs=None)
            tick_in.append(d)
            cur.update(d)
            U.copy_(d["U0"])
            launch(solver)
            torch.cuda.synchronize()
            last_host = i
            if t + 1 < args.warmup + args.steps:
                ts.advance(X.cpu().numpy(), U.cpu().numpy())
        del ts

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        best = step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    total_solves = B * world * args.steps
    value = total_solves / elapsed
    mean_iters = float(iters.float().mean().item())
    mean_passes = float(passes.float().mean().item())
    # linearisations executed: every pass is preceded by one, and a solve that stops on a rejection did one more
    st_h = status.cpu().numpy()
    mean_lin = mean_passes + float(np.mean(st_h != 0))

    out = None
    if rank == 0:
        bytes_launch = algorithmic_bytes_per_solve(N, M) * B
        if sampled:  # SURVEY §8(d) compact form: nominal 6·n_dyn·N + 3 offsets per sample instead of 6·M·N
            bytes_launch = (8 * (4 + 2 * N + 6 + 2 + 6 * n_dyn * N + 3 * n_dyn * n_smp) + 8 * (2 * N + 4 * (N + 1) + 1) + 8) * B
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        # F = I·N·(6·S + 100·M + 800) fp64 flops per solve, S = 200.  Priced twice: with I = the reference loop's
        # iteration count (what the CPU reference executes) and with the passes the kernel actually executes (it stops at the
        # first rejected iteration): linearisations·N·(6S + 100M) + passes·N·800.
        flops_ref = mean_iters * N * (6 * 200 + 100 * M + 800)
        flops_solve = mean_lin * N * (6 * 200 + 100 * M) + mean_passes * N * 800
        # which kernel family the library picks for this shape: asked, not restated (cilqr_solve_family)
        lanes = 64 if sampled else solver.solve_family(B, N, M)
        kernel_name = "cilqr_solve_kernel" if lanes == 64 else "cilqr_solve_groups_fast<%d>" % lanes
        if not sampled and lanes == 64 and solver.solve_wavefronts(B, N, M) >= 2:
            kernel_name = "cilqr_solve_share_kernel<%d>" % solver.solve_wavefronts(B, N, M)
        if sampled and solver.solve_sampled_wavefronts(B, N, n_dyn) > 1:
            kernel_name = "cilqr_solve_split_kernel<%d>" % solver.solve_sampled_wavefronts(B, N, n_dyn)
        traffic, traffic_tag = (None, None) if (args.workload == "c3" and args.materialised) else recorded_traffic(args.workload, B)
        out = {
            "metric": "CILQR solves/sec (N=%d, batch B)" % N, "value": value, "unit": "solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl + (", a closed-loop sequence of planner ticks (another batch every step; scenes.TickSequence)" if use_ticks else "")
                                   + ", inputs resident in HBM, + min-cost selection",
                       "batch_per_gpu": B, "horizon": N, "obstacles": M, "mean_reference_iterations": mean_iters,
                       "parallelism": "scene-sharded x%d, one ncclAllGather of 24 B per rank behind cilqr_argmin_global_device" % world},
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_tag,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "issue-bound path of one wavefront per SIMD: HBM fraction is tiny by design (LDS-resident solve); see fp64_valu"},
            "fp64_valu": {"achieved_tflops_est": flops_solve * B / (kern_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TF,
                          "frac": flops_solve * B / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                          "flops_per_solve_executed": flops_solve, "mean_executed_passes": mean_passes,
                          "mean_executed_linearisations": mean_lin,
                          "note": "algorithmic flops (SURVEY 8d) over the vector fp64 peak; the backward pass runs on v_mfma_f64_4x4x4 "
                                  "(matrix fp64 peak 157.3 TF), whose padded 4x4x4 blocks execute more flops than counted here",
                          "reference_iteration_figure": {"flops_per_solve": flops_ref,
                                                         "tflops": flops_ref * B / (kern_ms * 1e-3) / 1e12,
                                                         "note": "counts the rejected iterations the reference loop repeats and the kernel skips"}},
            "min_cost": {"J": best[0], "global_index": best[1]},
        }
        wave_family = lanes == 64
        if wave_family and B > 1024 and world == 1:
            # Batches beyond one solve per SIMD are dispatched longest-first by the pass counts of the PREVIOUS call (DESIGN.md
            # §4.1d).  That only pays when consecutive batches resemble each other solve by solve: beside the timed figure, the same
            # launches on a handle created with the hint switched off.
            os.environ["CILQR_NO_SCHEDULE_HINT"] = "1"
            plain = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=local_rank)
            del os.environ["CILQR_NO_SCHEDULE_HINT"]
            reps = list(range(args.warmup, args.warmup + min(args.steps, 20))) if use_ticks else [None] * 4
            c0 = [torch.cuda.Event(enable_timing=True) for _ in reps]
            c1 = [torch.cuda.Event(enable_timing=True) for _ in reps]
            for q, tk in enumerate(reps):
                if tk is not None:
                    cur.update(tick_in[tk])
                U.copy_(cur["U0"])
                c0[q].record()
                launch(plain)
                c1[q].record()
            torch.cuda.synchronize()
            skip = 0 if use_ticks else 1
            cold_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(c0[skip:], c1[skip:])]))
            plain.close()
            out["schedule_hint"] = {"active": True, "kernel_ms_without_hint": cold_ms, "solves_per_s_without_hint": B / (cold_ms * 1e-3),
                                    "note": ("the same ticks on a handle without the hint" if use_ticks else
                                             "value and roofline.kernel_ms are steady state on a REPEATED batch, the hint's best case: the "
                                             "solves are dispatched longest-first by the previous call's pass counts; without_hint = same "
                                             "launch, dispatch in index order")}
            if use_ticks:  # the repeated-batch figure, as a labelled note: the last tick solved five times over
                cur.update(tick_in[-1])  # (also what the parity figure below compares: U then holds the LAST tick's results)
                r0 = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                r1 = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                for q in range(5):
                    U.copy_(cur["U0"])
                    r0[q].record()
                    launch(solver)
                    r1[q].record()
                torch.cuda.synchronize()
                rep_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(r0[2:], r1[2:])]))
                out["schedule_hint"]["repeated_batch"] = {"kernel_ms": rep_ms, "solves_per_s": B / (rep_ms * 1e-3),
                                                          "note": "NOT the value: one batch solved again and again, where the previous call's "
                                                                  "pass counts predict this call's exactly (profiles/r03_schedule_hint_ticks.txt: "
                                                                  "on tick sequences the hint changes nothing)"}
        cnt = RECORDED["workloads"].get(args.workload, {}).get("counters", {})
        if traffic_tag and "SQ_WAVE_CYCLES" in cnt:  # SQ counter pass of the same command, per launch: recorded, not live
            out["issue"] = {"recorded_from": RECORDED_FROM, "recorded_head": RECORDED.get("recorded_head"),
                            "instruction_issue_cycles": cnt.get("SQ_ACTIVE_INST_ANY"), "wavefront_cycles": cnt["SQ_WAVE_CYCLES"],
                            "frac": cnt.get("SQ_ACTIVE_INST_ANY", 0.0) / cnt["SQ_WAVE_CYCLES"],
                            "wait_inst_any_frac": cnt.get("SQ_WAIT_INST_ANY", 0.0) / cnt["SQ_WAVE_CYCLES"],
                            "valu_insts": cnt.get("SQ_INSTS_VALU"), "mfma_f64_insts": cnt.get("SQ_INSTS_VALU_MFMA_F64"),
                            "salu_insts": cnt.get("SQ_INSTS_SALU"), "lds_insts": cnt.get("SQ_INSTS_LDS"),
                            "l2_hit_frac": (cnt["TCC_HIT_sum"] / cnt["TCC_REQ_sum"]) if cnt.get("TCC_REQ_sum") else None,
                            "note": "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES: share of resident-wavefront cycles in which an instruction "
                                    "issued (a lone wavefront issues one per 5.2 ticks, a matrix instruction 16.2)"}
        if args.workload == "c2" and B <= 1024 and world == 1 and not use_ticks:
            # A labelled note, never `value`: at one solve per SIMD a launch lasts as long as its slowest solve and most SIMDs idle
            # through its second half.  Do two INDEPENDENT batches in flight — two handles, two streams, as two
            # planners sharing the GPU would run — fill that tail?  (A single planner's consecutive ticks depend on each other.)
            other = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=local_rank)
            Ub, Xb, Jb = U0.clone(), torch.zeros_like(X), torch.zeros_like(J)
            itb, stb = torch.zeros_like(iters), torch.zeros_like(status)
            side = [torch.cuda.Stream(), torch.cuda.Stream()]
            lanes2 = ((solver, U, X, J, iters, status), (other, Ub, Xb, Jb, itb, stb))
            n2 = max(args.steps // 2, 5)

            def pair_round():
                for (slv, u, x, j, it_, st_), sd in zip(lanes2, side):
                    with torch.cuda.stream(sd):
                        u.copy_(U0)
                        slv.solve_batch_device(sd.cuda_stream, B, N, M, x0.data_ptr(), u.data_ptr(), poly.data_ptr(), xpl.data_ptr(),
                                               pose.data_ptr(), dim.data_ptr(), wts.data_ptr() if wts is not None else 0, x.data_ptr(),
                                               j.data_ptr(), it_.data_ptr(), st_.data_ptr())
            torch.cuda.synchronize()
            for _ in range(3):
                pair_round()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(n2):
                pair_round()
            torch.cuda.synchronize()
            tp = time.perf_counter() - tp
            same = bool(torch.equal(U, Ub) and torch.equal(iters, itb))
            other.close()
            out["two_batches_in_flight"] = {"value": 2 * n2 * B / tp, "unit": "solves/s", "ms_per_pair": 1e3 * tp / n2, "results_equal": same,
                                            "note": "NOT the value: two independent batches of %d solves on two handles and streams at once (no "
                                                    "min-cost selection) — how much of the tail that one launch leaves idle a second, "
                                                    "independent batch picks up (little: measured 2.45 M against 2.32 M, the launches "
                                                    "barely overlap; one launch of 2048 solves does: bench.py --batch 2048)" % B}
        # SURVEY §8(d) also asks for the host-buffer entry point (H2D + kernel + D2H); reported, never `value`
        reps = (5 if M <= 16 else 1) if world == 1 else 0
        if reps:
            def host_call(src, out=None):
                t = time.perf_counter()
                if sampled:
                    solver.solve_batch_sampled(N, src["x0"], src["U"], src["poly"], src["xplan_fl"], src["nom_pose"], src["nom_dim"],
                                               src["offsets"], sc["sample_weight"])
                else:
                    if out is not None:
                        out["U"][...] = src["U"]
                    solver.solve_batch(N, src["x0"], src["U"], src["poly"], src["xplan_fl"], src["obs_pose"], src["obs_dim"],
                                       src["obs_weight"], out=out)
                return time.perf_counter() - t
            host_call(sc)  # first call untimed
            hb = min(host_call(sc) for _ in range(reps))
            out["host_buffer_api"] = {"value": B / hb, "unit": "solves/s", "ms_per_batch": 1e3 * hb,
                                      "note": "cilqr_solve_batch from pageable host memory, PCIe copies included (best of %d)" % reps}
            if not sampled:  # the same call from page-locked buffers (cilqr_host_alloc): its copies are asynchronous DMA transfers
                pin = {k: (cilqr_amd.pinned_copy(sc[k]) if sc[k] is not None else None)
                       for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim", "obs_weight")}
                pout = dict(U=cilqr_amd.pinned_empty((B, 2 * N)), X=cilqr_amd.pinned_empty((B, 4 * (N + 1))), J=cilqr_amd.pinned_empty((B,)),
                            iters=cilqr_amd.pinned_empty((B,), np.int32), status=cilqr_amd.pinned_empty((B,), np.int32))
                host_call(pin, pout)
                hp = min(host_call(pin, pout) for _ in range(reps))
                out["host_buffer_api"]["pinned"] = {"value": B / hp, "unit": "solves/s", "ms_per_batch": 1e3 * hp,
                                                    "note": "the same call with every buffer from cilqr_host_alloc"}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 measurement
            from oracle import oracle as O
            O.build(ref=False)
            threads = O.max_threads()
            po = O.default_params(N)
            ns = min(B, 1024 if M <= 16 else 128)  # bounded sample
            sl = lambda a: None if a is None else np.ascontiguousarray(a[:ns])  # noqa: E731
            src = sc
            if use_ticks:  # the scenes of the LAST tick, whose results are the ones in U (materialised for the oracle where sampled)
                src = dict(last_host)
                if sampled:
                    src["obs_pose"], src["obs_dim"], src["obs_weight"] = scenes.materialise_samples(
                        last_host["nom_pose"][:ns], last_host["nom_dim"][:ns], last_host["offsets"][:ns], N)
            sample = [sl(src[k]) for k in ("x0", "U", "poly", "xplan_fl", "obs_pose", "obs_dim", "obs_weight")]
            res = {}

            def cpu_call():
                res["want"] = O.solve_batch(po, N, M, *sample, threads=threads)
            rate, spread, calls = cpu_rate(cpu_call, ns)
            want = res["want"]
            du = float(np.max(np.abs(U.cpu().numpy()[:ns] - want["U"])))
            out["cpu_baseline"] = {"value": rate, "unit": "solves/s", "cores": threads, "kind": "port",
                                   "sample": "the first %d scenes of the same batch (the last tick's), OpenMP schedule(dynamic) over the batch; one untimed "
                                             "call, then the median of 3 repetitions of %d calls (>= 2 s each)" % (ns, calls),
                                   "spread": spread, "per_core": rate / threads}
            out["max_abs_du_vs_oracle"] = du
            out["iters_equal_oracle"] = bool(np.array_equal(iters.cpu().numpy()[:ns], want["iters"]))
        emit(out)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
