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


# ---------------------------------------------------------------------------
# two REAL RCCL ranks, one per device (skipped on a one-GPU box; the driver's multi-GPU run has the devices)
# ---------------------------------------------------------------------------
def _device_count():
    import torch
    return torch.cuda.device_count()       # (counting devices does not initialise the GPU)


def _rccl_worker(rank, world, port, tmpdir):
    import ctypes
    import torch
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = str(rank)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        from sysbio_modeling_amd import _lib
        from sysbio_modeling_amd.distributed import evaluate_sharded, project_norms_evaluator
        # (1) the Python host's route: torch.distributed over RCCL
        proj, thetas = _build_project()
        norms, (lo, hi) = evaluate_sharded(project_norms_evaluator(proj), thetas, device=dev)
        np.save(os.path.join(tmpdir, 'rccl_norms_%d.npy' % rank), norms.cpu().numpy())
        # (2) the C ABI's route: sbm_allgather_norms on a communicator the host creates through RCCL's C API; the
        # unique id travels over the process group's store
        rccl = ctypes.CDLL('librccl.so.1')

        class UniqueId(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_char * 128)]
        uid = UniqueId()
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
            store.set('sbm_uid', bytes(uid))
        else:
            ctypes.memmove(ctypes.byref(uid), store.get('sbm_uid'), 128)
        comm = ctypes.c_void_p()
        rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
        try:
            ctx = proj._model.device_model.ctx
            n = 1000
            send = (torch.arange(n, dtype=torch.float64, device=dev) + 10000.0 * rank) * 0.5
            recv = torch.full((world * n,), -1.0, dtype=torch.float64, device=dev)
            _lib.check(ctx.lib.sbm_allgather_norms(ctx.handle, comm, _lib.dev_ptr(send), n, _lib.dev_ptr(recv)),
                       'sbm_allgather_norms')
            ctx.synchronize()
            torch.cuda.synchronize(dev)
            want = torch.cat([(torch.arange(n, dtype=torch.float64, device=dev) + 10000.0 * r) * 0.5 for r in range(world)])
            assert torch.equal(recv, want)
        finally:
            rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            rccl.ncclCommDestroy(comm)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_one_per_device(tmp_path):
    port = _free_port()
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / ('rccl_norms_%d.npy' % r)) for r in range(2))
    assert np.array_equal(a, b) and a.shape == (37,)
    proj, thetas = _build_project()
    assert np.array_equal(a, proj.evaluate_batch(thetas)['norms'])


def test_bench_multi_rank_line_on_one_gpu():
    """`python bench.py --gpus 2` end to end in rehearsal mode (SBM_BENCH_REHEARSAL=1: both ranks on GPU 0, gloo instead of
    RCCL -- RCCL refuses two ranks on one device): the launcher spawns its ranks itself, rank 0 prints ONE JSON line with
    the weak-scaling headline AND the configuration BASELINE.json names for the multi-GPU run, configs[3], sharded by
    vector over the ranks (strong scaling), every rank finding its own block in the gathered norms."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SBM_BENCH_REHEARSAL='1')
    p = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    assert len(lines[0]) < 4096                     # (the compact headline line; the full record is the side file)
    d = json.loads(lines[0])
    with open(os.path.join(repo, d['full'])) as fh:
        full = json.load(fh)
    assert full['value'] == pytest.approx(d['value'], rel=1e-5) and 'configs3_sharded' in full['configs']
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 1e8 and d['config']['failed_vectors'] == 0
    assert d['ranks']['world_size_seen'] == 2 and d['ranks']['gathered_norms_match_local_block']
    c3 = d['configs']['configs3_sharded']
    assert 'error' not in c3, c3
    assert c3['n_gpus'] == 2 and c3['vectors_per_rank'] == [512, 512] and c3['scaling'] == 'strong'
    assert c3['failed_vectors'] == 0 and c3['gathered_norms_match_local_block_on_every_rank'] and c3['value'] > 1e8
