#!/usr/bin/env python3
"""bench.py — CILQR solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: `cilqr_solve_batch_device` on BASELINE config 2 (B=1024 seeded synthetic
scenes, N=50, M=4 obstacles, fp64) with every input already resident in HBM, followed by the min-cost selection
(`cilqr_argmin_device`; with N > 1 ranks, one RCCL all-gather of the 16-byte (J, index) pairs — SURVEY §8e).  Each rank
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
FP64_VALU_PEAK_TF = 78.6   # vector fp64 = half the 157.3 TF fp32 vector peak of MI355X_MICROARCH.md


def algorithmic_bytes_per_solve(N, M):
    """bytes_in = 8(4 + 2N + 6 + 2 + 6MN), bytes_out = 8(2N + 4(N+1) + 1) + 8."""
    return 8 * (4 + 2 * N + 6 + 2 + 6 * M * N) + 8 * (2 * N + 4 * (N + 1) + 1) + 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="solves per GPU per step (BASELINE config 2: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import cilqr_amd
    from cilqr_amd import scenes
    from cilqr_amd.dist import select_min_cost

    B, N, M = args.batch, 50, 4
    p = cilqr_amd.default_params(N)
    sc = scenes.make_static(B, N, M, p, scenes.SEED0 + 2 + 1000 * rank)  # rank 0 == scenes.make_c2(B)
    solver = cilqr_amd.Solver(p, max_batch=B, max_horizon=N, max_obstacles=M, device=local_rank)
    dev = torch.device("cuda", local_rank)

    def dv(a, dtype=torch.float64):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(dev)
    x0, U0, poly, xpl = dv(sc["x0"]), dv(sc["U"]), dv(sc["poly"]), dv(sc["xplan_fl"])
    pose, dim = dv(sc["obs_pose"]), dv(sc["obs_dim"])
    U = U0.clone()
    X = torch.zeros(B, 4 * (N + 1), dtype=torch.float64, device=dev)
    J = torch.zeros(B, dtype=torch.float64, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    pair = torch.zeros(2, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(k=None):
        U.copy_(U0)
        if k is not None:
            ev0[k].record()
        solver.solve_batch_device(stream, B, N, M, x0.data_ptr(), U.data_ptr(), poly.data_ptr(), xpl.data_ptr(), pose.data_ptr(),
                                  dim.data_ptr(), 0, X.data_ptr(), J.data_ptr(), iters.data_ptr(), status.data_ptr())
        if k is not None:
            ev1[k].record()
        solver.argmin_device(stream, B, J.data_ptr(), pair.data_ptr())
        return select_min_cost(pair, rank * B, dist)  # all-gather of 16-byte pairs when world > 1

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

    out = None
    if rank == 0:
        bytes_launch = algorithmic_bytes_per_solve(N, M) * B
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        # fp64 VALU work actually needed per solve: accepted iterations k = (iters - 5)/2 for λ-exits
        flops_solve = mean_iters * N * (6 * 200 + 100 * M + 800)
        out = {
            "metric": "CILQR solves/sec (N=50, batch B)", "value": value, "unit": "solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: B=%d CILQR solves per GPU per step, N=50, M=4 static obstacles, "
                                   "inputs resident in HBM, + min-cost selection" % B,
                       "batch_per_gpu": B, "horizon": N, "obstacles": M, "mean_reference_iterations": mean_iters,
                       "parallelism": "scene-sharded x%d, RCCL all-gather of (J,index)" % world},
            "roofline": {"bound": "hbm", "kernel": "cilqr_solve_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "latency/fp64-VALU-bound path: HBM fraction is tiny by design (LDS-resident solve); see fp64_valu"},
            "fp64_valu": {"achieved_tflops_est": flops_solve * B / (kern_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TF,
                          "flops_per_solve_est": flops_solve},
            "min_cost": {"J": best[0], "global_index": best[1]},
        }
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            O.build(ref=False)
            threads = O.max_threads()
            po = O.default_params(N)
            t1 = time.perf_counter()
            want = O.solve_batch(po, N, M, sc["x0"], sc["U"], sc["poly"], sc["xplan_fl"], sc["obs_pose"], sc["obs_dim"], None, threads=threads)
            cpu_s = time.perf_counter() - t1
            du = float(np.max(np.abs(U.cpu().numpy() - want["U"])))
            out["cpu_baseline"] = {"value": B / cpu_s, "unit": "solves/s", "cores": threads, "kind": "port",
                                   "sample": "the same %d config-2 scenes, once, OpenMP over the batch" % B,
                                   "per_core": B / cpu_s / threads}
            out["max_abs_du_vs_oracle"] = du
            out["iters_equal_oracle"] = bool(np.array_equal(iters.cpu().numpy(), want["iters"]))
        print(json.dumps(out), flush=True)
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
