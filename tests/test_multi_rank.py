"""N > 1 path on CPU: world_size-2 gloo process group.  The batch shards by scene with no data-path collective; the one
exchange step is an all-gather of 24 bytes per rank: the (J_min, local index) pair and the rank's index offset (cilqr_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cases, out_q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "uncertainty-aware-cilqr-for-trajectory-optimization_amd"))
    from cilqr_amd.dist import select_min_cost
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for per_rank, shard in cases:
        pair = torch.tensor(per_rank[rank], dtype=torch.float64)
        res.append(select_min_cost(pair, rank * shard, dist))
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    # ([rank0 pair, rank1 pair], shard size)           expected (J, global index)
    (([3.5, 7.0], [2.25, 3.0]), 100),                  # rank 1 wins → 100 + 3
    (([1.0, 42.0], [1.0, 0.0]), 64),                   # tie on J → lowest GLOBAL index (rank 0's 42 < 64 + 0)
    (([float("inf"), -1.0], [9.0, 5.0]), 10),          # rank 0 had no finite cost
    (([float("nan"), 3.0], [float("inf"), -1.0]), 10),  # nothing usable anywhere
]
EXPECT = [(2.25, 103), (1.0, 42), (9.0, 15), (float("inf"), -1)]


def test_min_cost_selection_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, CASES, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        for (j, i), (ej, ei) in zip(got[r], EXPECT):
            assert i == ei and (j == ej or (j != j and ej != ej)), (r, j, i, ej, ei)
    assert got[0] == got[1]  # every rank learns the same winner


def test_single_rank_passthrough():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "uncertainty-aware-cilqr-for-trajectory-optimization_amd"))
    from cilqr_amd.dist import select_min_cost
    assert select_min_cost(torch.tensor([4.0, 9.0], dtype=torch.float64), 1000, None) == (4.0, 1009)
    assert select_min_cost(torch.tensor([float("inf"), -1.0], dtype=torch.float64), 1000, None) == (float("inf"), -1)


def test_rank_shards_are_disjoint_seeds(cilqr):
    """bench.py gives rank r the scenes of seed SEED0 + 2 + 1000 r: rank 0 is exactly BASELINE config 2, other ranks differ."""
    from cilqr_amd import scenes
    p = cilqr.default_params(50)
    a = scenes.make_c2(8, p)
    b = scenes.make_static(8, 50, 4, p, scenes.SEED0 + 2)
    c = scenes.make_static(8, 50, 4, p, scenes.SEED0 + 2 + 1000)
    assert np.array_equal(a["x0"], b["x0"]) and np.array_equal(a["obs_pose"], b["obs_pose"])
    assert not np.array_equal(a["x0"], c["x0"])


def test_shard_rule_covers_the_batch_contiguously_and_evenly(cilqr):
    """`cilqr_shard_range` (host arithmetic of cilqr_multi_solve_batch, no device needed): for every device count and the
    awkward batch sizes — empty, smaller than the device count, one off a multiple — the shards are contiguous, disjoint, cover
    [0, B) in device order and differ by at most one solve."""
    for n in (1, 2, 3, 8):
        sizes = {0, 1, n - 1, n, n + 1}
        for per in (1, 5, 128, 8192):
            sizes |= {n * per - 1, n * per, n * per + 1}
        for B in sorted(b for b in sizes if b >= 0):
            shards = [cilqr.shard_range(B, n, d) for d in range(n)]
            nxt = 0
            for first, count in shards:
                assert first == nxt and count >= 0, (n, B, shards)
                nxt = first + count
            assert nxt == B
            counts = [c for _, c in shards]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True), (n, B, counts)
            assert max(counts) == -(-B // n)  # what cilqr_create_multi's max_batch_per_device must hold
    for bad in ((-1, 2, 0), (4, 0, 0), (4, 2, 2), (4, 2, -1)):
        with pytest.raises(cilqr.CilqrError):
            cilqr.shard_range(*bad)
