"""Multi-GPU plumbing of the hot path: one process per GPU, sequences sharded by whole records,
ONE exchange step -- an all-reduce(sum) of {count table, ltot, 84 background counters} -- after which
every rank holds the global tables and the pattern-space sweeps / EM partition trivially.

The non-overlap rule of src/base_pattern.cpp:361-366 never reaches across a sequence boundary
(:382 `j += pattern_length`), so per-shard counts add exactly.  The background MODEL is
additive in its 84 counters, not in V, so V is derived after the reduction.

torch.distributed is the transport: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import torch


def shard_range(n_seq, rank, world):
    """Contiguous whole-sequence shard [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_seq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def split_round_robin(n, rank, world):
    """Indices of the PWMs (or IUPAC mutants) rank works on after the reduction."""
    return list(range(rank, n, world))


def check_global_bin_bound(local_bound, dist=None):
    """The count table travels as 32-bit words; sums stay exact while the GLOBAL bound on a single bin
    (sum over runs of ceil(windows / W), pengk_packed.max_bin_bound) is below 2^32."""
    t = torch.tensor([int(local_bound)], dtype=torch.int64)
    if dist is not None and dist.is_initialized():
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        t = t.cpu()
    total = int(t.item())
    if total >= 2 ** 32:
        raise OverflowError("a count bin could reach %d >= 2^32 across ranks; use fewer sequences per job" % total)
    return total


def allreduce_tables(counts_i32, scalars_i64, dist=None):
    """In-place sum over ranks.  counts_i32: int32 view of the uint32 count table (two's-complement adds are
    the uint32 adds); scalars_i64: int64[85] = 84 background counters + ltot."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    assert counts_i32.dtype == torch.int32 and scalars_i64.dtype == torch.int64
    if dist.get_backend() == "gloo" and counts_i32.is_cuda:  # rehearsal path: gloo reduces host copies
        c, s = counts_i32.cpu(), scalars_i64.cpu()
        dist.all_reduce(c)
        dist.all_reduce(s)
        counts_i32.copy_(c)
        scalars_i64.copy_(s)
        return
    dist.all_reduce(counts_i32)
    dist.all_reduce(scalars_i64)
