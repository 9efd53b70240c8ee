"""N > 1 path on CPU: world_size-2 gloo processes, each counting its whole-sequence shard (the oracle
stands in for the device kernels here), then the ONE all-reduce the GPU path performs.  Checks shard
additivity of counts / ltot / background counters and the reduce plumbing of peng-motif_amd/sharding.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fasta, W, both, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as po
    import peng_motif_amd as pk
    from peng_motif_amd import sharding

    codes, offs = po.read_fasta(fasta)
    n = len(offs) - 1
    lo, hi = sharding.shard_range(n, rank, world)
    sc, so = codes[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo]
    packed = pk.Packed(sc, so, W)  # host packer runs per shard, as on the GPU path
    counts, ltot = po.count(sc, so, W, both)
    if both:  # the device count leaves canonical bins only; mirror after the reduction
        canon = np.array([x <= po.revcomp(x, W) for x in range(4 ** W)])
        counts = np.where(canon, counts, 0)
    assert packed.n_windows == ltot
    sharding.check_global_bin_bound(packed.max_bin_bound, dist)
    c32 = torch.from_numpy(counts.astype(np.uint32).view(np.int32).copy())
    scal = torch.from_numpy(np.concatenate([packed.bg_counts, [ltot]]).astype(np.int64))
    sharding.allreduce_tables(c32, scal, dist)
    np.save(os.path.join(out_dir, "counts%d.npy" % rank), c32.numpy().view(np.uint32))
    np.save(os.path.join(out_dir, "scal%d.npy" % rank), scal.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("fasta,W,both", [("MafK_100seqs.fasta", 8, True), ("torture.fa", 6, False)])
def test_two_rank_allreduce_matches_single_rank(golden_dir, tmp_path, fasta, W, both):
    from oracle import oracle as po
    path = os.path.join(golden_dir, fasta)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), path, W, both, str(tmp_path)), nprocs=world, join=True)
    codes, offs = po.read_fasta(path)
    want, ltot = po.count(codes, offs, W, both)
    bg = po.bg_counts(codes, offs, 2)
    for r in range(world):
        got = np.load(tmp_path / ("counts%d.npy" % r)).astype(np.uint64)
        if both:  # mirror (src/base_pattern.cpp:387-392) commutes with the sum
            for x in range(4 ** W):
                rc = po.revcomp(x, W)
                if x > rc:
                    got[x] = got[rc]
        assert np.array_equal(got, want)
        scal = np.load(tmp_path / ("scal%d.npy" % r))
        assert int(scal[84]) == ltot and np.array_equal(scal[:84], bg)


def test_shard_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    from peng_motif_amd import sharding
    for n in (0, 1, 7, 8, 25, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.split_round_robin(10, 1, 4) == [1, 5, 9]
    with pytest.raises(OverflowError):
        sharding.check_global_bin_bound(2 ** 32)
