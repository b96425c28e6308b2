"""N > 1 path on CPU: world size 2, gloo.  Covers the vector-axis sharding and the gather of
per-vector norms that RCCL carries on the GPUs (the integrator itself needs no collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sysbio_modeling_amd.distributed import shard_range, gather_norms, evaluate_sharded


def test_shard_range_partitions_the_vector_axis():
    for V in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(V, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == V
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, V, q, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        thetas = rng.standard_normal((V, q))          # same on every rank

        def fake_norms(block):                         # stands in for Project.evaluate_batch(...)['norms']
            return torch.from_numpy((block ** 2).sum(axis=1))
        norms, (lo, hi) = evaluate_sharded(fake_norms, thetas)
        want = (thetas ** 2).sum(axis=1)
        assert norms.shape == (V,)
        assert np.array_equal(norms.numpy(), want)     # bit-exact: a gather moves, it does not add
        assert (lo, hi) == shard_range(V, rank, world)
        # unknown global size: counts are exchanged first
        again = gather_norms(torch.from_numpy(want[lo:hi]))
        assert np.array_equal(again.numpy(), want)
        best = int(torch.argmin(norms))
        assert best == int(np.argmin(want))
        with open(os.path.join(tmpdir, 'ok_%d' % rank), 'w') as fh:
            fh.write('%d %d' % (lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('V', [10, 11])
def test_gloo_world2_sharded_norms(tmp_path, V):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, V, 5, str(tmp_path)), nprocs=2, join=True)
    spans = [tuple(int(x) for x in open(tmp_path / ('ok_%d' % r)).read().split()) for r in range(2)]
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == V


def test_single_process_is_a_no_op():
    x = torch.arange(5, dtype=torch.float64)
    assert gather_norms(x) is x
    norms, span = evaluate_sharded(lambda b: torch.from_numpy(b.sum(axis=1)), np.ones((4, 3)))
    assert span == (0, 4) and norms.tolist() == [3.0] * 4
