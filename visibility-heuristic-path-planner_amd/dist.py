"""Sharding of independent light sources over the GPUs of one node (torch.distributed / RCCL).

The sweep has no exchange step -- every source only reads the shared occupancy map -- so
sources are block-partitioned over ranks and each rank sweeps its own shard with no
collective on the data path.  What a caller does with the per-source fields decides the
collective afterwards:

  gather_fields  : every rank gets every field (RCCL all-gather over xGMI).  Volume is
                   world * S_local * nx*ny * sizeof(elem) per rank: for BASELINE config 5
                   (4096^2, 1024 sources, fp64) that is 137 GB landed on every GPU, far more
                   time than the sweeps themselves -- use fp32 fields or the union below.
  union_fields   : what a multi-source planner needs: per cell the best visibility over all
                   sources and which source gave it (max-union + arg-source).  One all-gather
                   of world partial unions (nx*ny elements each) instead of all fields.

The planner's pivot sequence itself does not shard (pivot k+1 depends on the union after
pivot k): run replicas, one planner per GPU.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_sources, rank, world):
    """[lo, hi) of the contiguous block of sources rank `rank` owns (sizes differ by at most 1)."""
    base, extra = divmod(n_sources, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sources(sources, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(len(sources), rank, world)
    return sources[lo:hi], lo


def broadcast_map(occ, src=0):
    """The occupancy map (uint8 [ny, nx] tensor on this rank's device) from rank `src` to every rank: one RCCL
    broadcast of nx*ny bytes (16.8 MB at 4096^2), SURVEY 8(e).  Returns the tensor (filled in place)."""
    dist.broadcast(occ, src=src)
    return occ


def gather_fields(local_fields, n_sources, out=None):
    """local_fields [n_local, ny, nx] on this rank's device -> [n_sources, ny, nx] on every rank.

    Equal shards (n_sources % world == 0, the BASELINE configurations): ONE all-gather straight into `out` (allocated
    here if not given) -- no padding, no concatenation, no second copy of the result (at config 5 in fp64 the result
    alone is 137 GB of the 288 GB).  Ragged shards are padded to the largest and trimmed."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_sources % world == 0:
        assert local_fields.shape[0] == n_sources // world
        if out is None:
            out = local_fields.new_empty((n_sources,) + tuple(local_fields.shape[1:]))
        dist.all_gather_into_tensor(out, local_fields.contiguous())
        return out
    cap = (n_sources + world - 1) // world  # all_gather_into_tensor needs equal shards: pad to the largest
    pad = local_fields.new_zeros((cap,) + tuple(local_fields.shape[1:]))
    pad[: local_fields.shape[0]] = local_fields
    out = local_fields.new_empty((world * cap,) + tuple(local_fields.shape[1:]))
    dist.all_gather_into_tensor(out, pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(n_sources, r, world)
        pieces.append(out[r * cap: r * cap + (hi - lo)])
    return torch.cat(pieces, 0)


def _union_shim(fields, labels, first_index):
    """The reduction of vhp_union_fields_device / vhp_union_partials_device (csrc/vhp_union.hip.h) in torch ops, for tensors that are
    not on a GPU (the gloo tests): one pass, field after field, replace on strict improvement -- or, with label fields, on a tie
    with a lower label --, no temporaries beyond one mask.  Returns (best, arg int32); no field at all: (-1, INT32_MAX)."""
    shape = tuple(fields.shape[1:])
    best = fields.new_full(shape, -1.0)
    arg = torch.full(shape, 0x7fffffff, dtype=torch.int32, device=fields.device)
    for k in range(fields.shape[0]):
        f = fields[k]
        lab = labels[k] if labels is not None else None
        better = f > best
        if lab is not None:
            better |= (f == best) & (lab < arg)
        best = torch.where(better, f, best)
        arg = torch.where(better, lab if lab is not None else torch.full_like(arg, first_index + k), arg)
    return best, arg


def union_fields(local_fields, first_index, n_sources, ctx=None):
    """Max-union over ALL sources and the index of the (first) source attaining it.

    Returns (best [ny, nx], arg_source int64 [ny, nx]); ties resolve to the lowest source index, like a sequential max-union
    that only replaces on strict improvement (reference src/visibilityBasedSolver.cpp:417-418).  Every rank reduces its own
    shard in ONE pass over its fields (ctx given and the fields on its GPU: the HIP kernel behind vhp_union_fields_device --
    no temporaries; at BASELINE config 5 the torch expression this replaces made a 17 GB int64 temporary and read the 17 GB of
    fields four times), the partials -- one union field and one label field per rank -- are all-gathered (RCCL over xGMI, or
    gloo), and every rank merges them (vhp_union_partials_device: a tie goes to the lowest label)."""
    world = dist.get_world_size()
    on_gpu = ctx is not None and local_fields.is_cuda
    shape = tuple(local_fields.shape[1:])
    if on_gpu:
        import vhp_amd
        vdt = vhp_amd.F64 if local_fields.dtype == torch.float64 else vhp_amd.F32
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        local_fields = local_fields.contiguous()
        best = local_fields.new_empty(shape)
        arg = torch.empty(shape, dtype=torch.int32, device=local_fields.device)
        ctx.union_fields_device(local_fields.data_ptr() if local_fields.shape[0] else best.data_ptr(), local_fields.shape[0], best.data_ptr(), arg.data_ptr(),
                                first_index=first_index, dtype=vdt)
    else:
        best, arg = _union_shim(local_fields, None, first_index)
    # concatenated-along-dim-0 form: accepted by both RCCL and gloo
    all_best = best.new_empty((world,) + shape)
    all_arg = arg.new_empty((world,) + shape)
    dist.all_gather_into_tensor(all_best.view((world * shape[0],) + shape[1:]), best.contiguous())
    dist.all_gather_into_tensor(all_arg.view((world * shape[0],) + shape[1:]), arg.contiguous())
    if on_gpu:
        gbest = best.new_empty(shape)
        garg = arg.new_empty(shape)
        ctx.union_partials_device(all_best.data_ptr(), all_arg.data_ptr(), world, gbest.data_ptr(), garg.data_ptr(), dtype=vdt)
    else:
        gbest, garg = _union_shim(all_best, all_arg, 0)
    # (a cell that no field covers -- no sources at all -- is labelled n_sources, as before)
    return gbest, torch.clamp(garg.to(torch.int64), max=n_sources)


def sweep_gather_overlapped(launch, sources, out, chunks=4):
    """Strong-scaling form of "sweep my shard, all-gather the fields": the shard is cut into `chunks` pieces; while the
    sweeps of piece k+1 run on the compute stream, the all-gather of piece k runs on a second stream (RCCL over xGMI is
    per-link bound and far slower than the sweeps, SURVEY 8(e): overlapping hides the compute, not the wire time).

    launch(lo, hi, dst) must enqueue the sweeps of this rank's shard sources [lo, hi) into dst ([hi-lo, ny, nx], a
    view of this rank's block of `out`) on the CURRENT stream and return without synchronising.
    out: [n_sources, ny, nx], n_sources % (world * chunks) == 0; on return (after the final synchronisation by the
    caller) piece k of rank r sits at out[k * world * m + r * m : ... + m], m = n_sources / (world * chunks) -- the
    layout an all-gather of equal pieces gives; `gathered_index` maps a source index to its row."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = len(sources)
    assert n % (world * chunks) == 0
    per_rank = n // world
    m = per_rank // chunks
    cuda = out.is_cuda
    comm = torch.cuda.Stream() if cuda else None
    works = []
    for k in range(chunks):
        block = out[k * world * m: (k + 1) * world * m]
        mine = block[rank * m: (rank + 1) * m]
        launch(k * m, (k + 1) * m, mine)
        if cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                works.append(dist.all_gather_into_tensor(block, mine, async_op=True))
        else:
            works.append(dist.all_gather_into_tensor(block, mine.clone(), async_op=True))
    for w in works:
        w.wait()
    if cuda:
        torch.cuda.current_stream().wait_stream(comm)
    return out


def gathered_index(source_index, n_sources, world, chunks):
    """Row of `out` (sweep_gather_overlapped) that holds the field of global source `source_index`, sources being
    block-partitioned over ranks (shard_bounds) and each shard cut into `chunks` pieces."""
    per_rank = n_sources // world
    m = per_rank // chunks
    r, local = divmod(source_index, per_rank)
    k, j = divmod(local, m)
    return k * world * m + r * m + j


def sweep_sharded(compute, sources, mode="none", ctx=None):
    """compute(shard [n,2] int32) -> fields [n, ny, nx] tensor on this rank's device.

    mode: "none" -> (local fields, first index); "gather" -> all fields; "union" -> (best, arg_source) (ctx: the vhp context of this
    rank's device -- the reduction then runs as the HIP kernel of vhp_union_fields_device, see union_fields)."""
    shard, lo = shard_sources(sources)
    local = compute(shard)
    if mode == "none":
        return local, lo
    if mode == "gather":
        return gather_fields(local, len(sources))
    if mode == "union":
        return union_fields(local, lo, len(sources), ctx=ctx)
    raise ValueError(mode)
