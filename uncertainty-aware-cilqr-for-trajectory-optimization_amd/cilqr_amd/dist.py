"""Cross-rank min-cost selection: the one exchange step of the sharded batch.

Every rank solves its own shard of scenes with no data-path collective.  Afterwards each rank holds one 16-byte pair
(J_min, local index) from `cilqr_argmin_device`; ONE all-gather of those pairs, each with its rank's index offset appended, (RCCL over xGMI when the process group is
"nccl"; gloo in the CPU tests) followed by a local lexicographic minimum gives every rank the same global winner, with the
lowest global index winning ties (the strict-< first-minimum convention of the reference's argmins,
I/Constraints.cpp:50).  The message is 24 B per rank — latency-bound, so ring vs tree and xGMI link bandwidth are
irrelevant.
"""
import torch

_offset_cache = {}


def select_min_cost(pair, index_offset, dist=None, group=None):
    """pair: tensor [J_min, local_index] (float64, on the rank's device); index_offset: global index of this rank's first
    scene.  Returns (J_min, global_index) as Python numbers, identical on every rank.  index -1 ⇒ no finite cost."""
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
