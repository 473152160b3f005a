"""Rank sharding of the inference path ("256 sequences -> 32/GPU, no collective; gather of the
(256, steps, 225) outputs at the end").  The reference gets this from tf.distribute: `strategy.run(step_fn,
args=(next(iterator),))` in mint/ctl/single_task_evaluator.py:67-86 hands every replica its slice of the global
batch, and infer_auto_regressive (mint/core/fact_model.py:103-132) treats sequences independently.  Here: one
process per GPU, rank r keeps rows [r*B/N, (r+1)*B/N) of every per-sample entry of the input dict; nothing is
exchanged while sequences are generated; `gather_rows` collects the per-rank outputs on every rank at the end."""
import torch


def world_info():
    """(rank, world_size) of the default process group; (0, 1) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total, rank, world):
    """Rows [lo, hi) of `total` that rank `rank` of `world` owns: contiguous, sizes differ by at most one, earlier
    ranks take the larger shares (numpy.array_split's rule), every row owned exactly once."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_inputs(inputs, rank=None, world=None):
    """This rank's slice of a batch dict: every entry whose leading length equals the batch size (tensors, arrays,
    lists of names) is cut to shard_range; anything else (scalars, config entries) is passed through."""
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return inputs
    batch = int(inputs["motion_input"].shape[0])
    lo, hi = shard_range(batch, rank, world)
    out = {}
    for k, v in inputs.items():
        n = v.shape[0] if hasattr(v, "shape") and len(getattr(v, "shape", ())) > 0 else (
            len(v) if isinstance(v, (list, tuple)) else None)
        out[k] = v[lo:hi] if n == batch else v
    return out


def gather_rows(local, total, rank=None, world=None, group=None):
    """All ranks' row blocks (shard_range order) concatenated along dim 0 -> (total, ...) on every rank.  One
    all_gather of equal-sized (padded) blocks after generation: the only communication of the inference path.
    RCCL (backend "nccl") gathers device tensors in place; gloo (CPU tests, dry runs) goes through the host."""
    import torch.distributed as dist
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    most = max(hi - lo for lo, hi in sizes)
    if int(local.shape[0]) != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("rank %d holds %d rows, its shard of %d has %d" % (
            rank, local.shape[0], total, sizes[rank][1] - sizes[rank][0]))
    host = dist.get_backend(group) == "gloo"
    src = local.detach().cpu() if host else local.detach()
    pad = src.new_zeros((most,) + tuple(src.shape[1:]))
    pad[: src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    full = torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
    return full.to(local.device) if host else full
