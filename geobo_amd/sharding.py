"""Voxel-column sharding of the inversion across the GPUs of one node (one process per GPU).

The path shards by voxel COLUMNS of AK / V: once L is known every column is
independent.  The only data-path collectives are
    all-reduce(sum)  of the M_pad x M_pad partial AkA   (each rank contracts its own columns)
    all-gather       of the per-rank mu / var slices
issued through torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
Everything here is device agnostic so the N > 1 host logic is covered by world_size-2 gloo tests.
"""
import numpy as np
import torch

PAD_N = 128


class EmulatedGroup:
    """Stand-in for a process group: ONE process plays rank `rank` of `world` alone on its device (tools/emulate_rank.py: what
    does a rank's compute of the sharded step cost, measured where no multi-GPU node is available).  Every collective below becomes
    its local part only: an all-reduce is the identity, an all-gather / all-to-all returns this rank's own contribution in every
    slot (right sizes, finite values, wrong numbers for the peers' parts -- the caller injects what it needs to stay well posed)."""

    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)


def backend_of(group):
    """'emulate' for an EmulatedGroup, else the torch.distributed backend name of the group (None: not initialised)."""
    if isinstance(group, EmulatedGroup):
        return "emulate"
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return None
    return torch.distributed.get_backend(group)


def shard_columns(n_pad, world, rank):
    """Contiguous voxel-column range [c0, c1) of `rank`, in units of 128 columns."""
    units = n_pad // PAD_N
    u0 = units * rank // world
    u1 = units * (rank + 1) // world
    return u0 * PAD_N, u1 * PAD_N


def _live(world, group, force):
    """Does a collective go to the backend?  Not under an EmulatedGroup; with one rank only when `force` is set AND a process group
    exists (bench.py --gpus 1 --check and the `-m gpu` RCCL test: the same calls a rank of N > 1 makes, issued through RCCL with one
    rank, so that library load, fp64 support and stream semantics are exercised on a one-GPU box)."""
    if isinstance(group, EmulatedGroup):
        return False
    if world > 1:
        return True
    return bool(force) and torch.distributed.is_available() and torch.distributed.is_initialized()


def allreduce_sum_(t, world, group=None, force=False):
    """In-place sum over ranks (partial AkA of the column form; partial sums of squares of the row form)."""
    if _live(world, group, force):
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


def agree(flag, world, group=None, device=None, force=False):
    """Every rank must take the same form of a step (a rank-divergent decision would deadlock in the first collective).  ONE
    all-reduce (MIN over [flag, -flag] gives the minimum and minus the maximum) and one read-back; raises when ranks differ."""
    if not _live(world, group, force):
        return flag
    v = 1 if flag else 0
    t = torch.tensor([v, -v], dtype=torch.int32, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=group)
    lo, neg_hi = (int(x) for x in t.tolist())
    if lo != -neg_hi:
        raise RuntimeError("ranks disagree on the form of this step (this rank: %s): operators were built differently" % bool(flag))
    return flag


def gather_slices(t, nblocks, n_pad, world, group=None):
    """All-gather per-rank vectors of length nblocks * (c1 - c0); shard sizes may differ by one 128-unit, so the
    payload is padded to the largest shard.  Returns the list of per-rank tensors (trimmed)."""
    if world == 1:
        return [t]
    sizes = []
    for r in range(world):
        c0, c1 = shard_columns(n_pad, world, r)
        sizes.append((c1 - c0) * nblocks)
    mx = max(sizes)
    buf = torch.zeros(mx, dtype=t.dtype, device=t.device)
    buf[:t.numel()] = t
    if isinstance(group, EmulatedGroup):
        return [buf[:n] for n in sizes]
    outs = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
    torch.distributed.all_gather(outs, buf, group=group)
    return [o[:n] for o, n in zip(outs, sizes)]


def gather_rows(local, world, group=None, force=False):
    """All-gather of equal row blocks: `local` (any shape, contiguous) of every rank -> tensor (world, *local.shape).
    Used by the row-sharded lattice Gram: every rank correlates its own sensor rows of A K with the stencil table, so AkA
    arrives as row blocks (0.57 GB in total at 64^3) instead of as partial sums that need an all-reduce."""
    if world == 1 and not _live(world, group, force):
        return local.unsqueeze(0)
    import torch.distributed as dist
    local = local.contiguous()
    if isinstance(group, EmulatedGroup):
        return local.unsqueeze(0).expand((world,) + tuple(local.shape))
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=group)
    else:
        dist.all_gather([out[r] for r in range(world)], local, group=group)
    return out


def assemble_columns(parts, props, n, n_pad, world, to_host=None):
    """Per-rank [P_c x ncols_r] slices -> one (3n,) host vector in the reference's property-major order
    (NaN for property blocks that were not computed).  to_host(tensor, slot) -> 1-D host array that stays valid until the next
    call with the same slot (the engine's pinned staging buffers); default: a pageable copy per part."""
    out = np.full(3 * n, np.nan)
    if to_host is not None and len(parts) > 1 and all(isinstance(p, torch.Tensor) and p.is_cuda for p in parts):
        # one device-to-host copy for all ranks' slices (8 per-part copies, each with its own stream sync, cost 2 ms at 8 ranks)
        sizes = [int(p.numel()) for p in parts]
        flat = to_host(torch.cat([p.reshape(-1) for p in parts]), 0)
        offs = np.cumsum([0] + sizes)
        parts = [flat[offs[i]:offs[i + 1]] for i in range(len(sizes))]
    for r, part in enumerate(parts):
        c0, c1 = shard_columns(n_pad, world, r)
        ncr = c1 - c0
        hi = min(c1, n)
        if hi <= c0:
            continue
        if isinstance(part, torch.Tensor):
            ph = to_host(part, r) if (to_host is not None and part.is_cuda) else part.detach().cpu().numpy()
        else:
            ph = np.asarray(part)
        for jj, j in enumerate(props):
            out[j * n + c0:j * n + hi] = ph[jj * ncr:jj * ncr + (hi - c0)]
    return out


def exchange_blocks(send, world, group=None):
    """All-to-all of equal blocks: `send` is (world, blk) contiguous, row d goes to rank d; returns (world, blk) whose row s
    came from rank s.  RCCL: one all_to_all_single over xGMI (the exchange step of the row-sharded spectral product: every
    rank transforms its own sensor rows and hands each peer the block-columns that peer owns).  Backends without
    all-to-all (gloo: CPU tests, single-GPU dry runs) fall back to an all-gather + selection."""
    if world == 1 or isinstance(group, EmulatedGroup):
        return send
    import torch.distributed as dist
    assert send.dim() == 2 and send.shape[0] == world and send.is_contiguous()
    recv = torch.empty_like(send)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(recv, send, group=group)
    else:
        # emulation: one all-gather per destination (a buffer of one send size at a time instead of world of them)
        me = dist.get_rank(group)
        parts = [torch.empty_like(send[0]) for _ in range(world)]
        for dest in range(world):
            dist.all_gather(parts, send[dest].contiguous(), group=group)
            if dest == me:
                for src in range(world):
                    recv[src].copy_(parts[src])
    return recv


def exchange_blocks_start(send, world, group=None, out=None):
    """exchange_blocks in two halves, so that a caller with more to compute can put that between them: on RCCL the all-to-all is
    issued asynchronously (it runs on the communicator's stream, ordered after everything queued on the current stream so far) and
    (recv, work) is returned -- `exchange_blocks_finish(work)` makes the current stream wait for it.  Other backends (gloo: CPU
    tests, single-GPU dry runs) do the whole exchange here and return work = None."""
    if world == 1 or isinstance(group, EmulatedGroup):
        return send, None
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        assert send.dim() == 2 and send.shape[0] == world and send.is_contiguous()
        recv = out if out is not None else torch.empty_like(send)
        return recv, dist.all_to_all_single(recv, send, group=group, async_op=True)
    return exchange_blocks(send, world, group), None


def exchange_blocks_finish(work):
    if work is not None:
        work.wait()
