"""Cross-rank min-cost selection: the one exchange step of the sharded batch.

Every rank solves its own shard of scenes with no data-path collective.  The exchange itself lives behind the C-ABI
(`cilqr_argmin_global_device`, csrc/cilqr_comm.cpp): per-rank argmin → ONE `ncclAllGather` of a 24-byte record
{J_min, local index, index offset} per rank (RCCL over xGMI) → lexicographic minimum on the device, lowest global index
winning ties (the strict-< first-minimum convention of the reference's argmins, I/Constraints.cpp:50).  The message is
bytes — latency-bound, so ring vs tree and xGMI link bandwidth are irrelevant.

This module is the torch.distributed side of it:
  * `init_comm(solver, dist)`  one process per GPU: rank 0 makes the RCCL unique id through the C-ABI, torch.distributed
    carries the 128 bytes to the other ranks, every rank joins the handle's communicator;
  * `select_min_cost_device`    the step itself (C-ABI) + the one 16-byte device→host read;
  * `select_min_cost`           the same combine rule over a torch process group WITHOUT RCCL (gloo on CPU): what the
    world-size-2 CPU tests run, since RCCL needs GPUs.
"""
import torch

_offset_cache = {}


def init_comm(solver, dist, group=None):
    """Joins `solver`'s handle to an RCCL communicator spanning the ranks of the (initialised) torch process group."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    from . import comm_unique_id
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    solver.comm_init_rank(world, rank, box[0])  # ncclCommInitRank: collective over the ranks
    return world


def select_min_cost_device(solver, stream, B, J_ptr, index_offset, pair):
    """`cilqr_argmin_global_device` on `stream`, then the one host read.  pair: 2-double device tensor (scratch/out).
    Returns (J_min, global_index), identical on every rank; index -1 ⇒ no finite cost anywhere."""
    solver.argmin_global_device(stream, B, J_ptr, index_offset, pair.data_ptr())
    j, i = pair.tolist()
    return j, int(i)


def select_min_cost(pair, index_offset, dist=None, group=None):
    """pair: tensor [J_min, local_index] (float64, from `cilqr_argmin_device`); index_offset: global index of this rank's
    first scene.  The combine rule of `cilqr_argmin_global_device` over a torch process group (gloo in the CPU tests).
    Returns (J_min, global_index) as Python numbers, identical on every rank.  index -1 ⇒ no finite cost."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        j, i = pair.tolist()  # the one device→host read of the step
        return j, (int(i) + index_offset if i >= 0 else int(i))
    world = dist.get_world_size(group)
    # the pair travels with its rank's offset (24 B per rank) so that the global index is formed after the one host read
    key = (str(pair.device), pair.dtype, int(index_offset))
    if key not in _offset_cache:  # uploaded once per (device, offset), not once per step
        _offset_cache[key] = torch.tensor([float(index_offset)], dtype=pair.dtype, device=pair.device)
    mine = torch.cat([pair.reshape(2), _offset_cache[key]])
    allp = torch.empty(world, 3, dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(allp, mine.reshape(1, 3), group=group)
    rows = [(j, i + off if i >= 0 else i) for j, i, off in allp.tolist()]
    best = None
    for j, i in rows:
        if i < 0 or j != j:
            continue
        if best is None or j < best[0] or (j == best[0] and i < best[1]):
            best = (j, int(i))
    return best if best is not None else (float("inf"), -1)
