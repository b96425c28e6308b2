"""Two ranks, real kernels: each process evaluates its block of the ensemble on the GPU, the residual norms
are gathered across processes (gloo here: both ranks share the box's one GPU, which RCCL refuses; the
driver's multi-GPU run uses backend 'nccl' through the same code) and must equal the single-process result
bitwise -- the path has no data-path collective, so sharding cannot change a number."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_project():
    import warnings
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('cascade20')
    model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(model, n_exp=2)
    return proj, models_zoo.config4_ensemble(th0, 37, spread=0.3)


def _worker(rank, world, port, tmpdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = '0'                      # one GPU on this box: both ranks use it
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sysbio_modeling_amd.distributed import evaluate_sharded, project_norms_evaluator, shard_range
        proj, thetas = _build_project()
        gpu_eval = project_norms_evaluator(proj)
        norms, (lo, hi) = evaluate_sharded(lambda block: gpu_eval(block).cpu(), thetas)
        assert (lo, hi) == shard_range(len(thetas), rank, world)
        np.save(os.path.join(tmpdir, 'norms_%d.npy' % rank), norms.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / ('norms_%d.npy' % r)) for r in range(2))
    assert np.array_equal(a, b) and a.shape == (37,)
    proj, thetas = _build_project()
    single = proj.evaluate_batch(thetas)['norms']
    assert np.array_equal(a, single)
