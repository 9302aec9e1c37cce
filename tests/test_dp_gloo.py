"""world_size-2 `gloo` tests (CPU) of the data-parallel path's host logic and semantics:
  * each rank takes rows [r*B, (r+1)*B) of the global batch (dataset sharding);
  * mean of the per-rank gradients (what the RCCL sum all-reduce followed by the 1/N scale in the Adam kernel computes)
    equals the gradient of the global batch -- checked with the oracle as the per-rank gradient engine;
  * the 128-byte communicator id travels by broadcast_object_list (the bootstrap bench.py / the CLI use).
The RCCL collective itself needs GPUs: tests/test_gpu_model.py runs the same call sequence with a 1-rank communicator."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import transformer_oracle as O
from composer_amd import dataset as D

V, E, H, L, W, T, B = 390, 32, 4, 1, 16, 16, 2


def _worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = O.init_params(V, E, W, L, seed=4)
        ds = D.load_dataset(D.get_processed_files(tmp), B, T, shuffle=True, seed=3, rank=rank, world_size=world)
        x, y = next(iter(ds))
        orc = O.OracleTransformer(O.Config(V, E, W, L, H), params)
        loss, acc, G, _ = orc.loss_and_grads(x, y, training=False)
        flat = torch.tensor(np.concatenate([G[k].ravel() for k in sorted(G)]))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world                                               # grad_scale = 1/nranks (model.hip: adam())
        met = torch.tensor([loss, acc]); dist.all_reduce(met); met /= world
        uid = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        q.put((rank, x, y, flat.numpy(), met.numpy(), uid[0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_mean_equals_global_batch(tmp_path):
    D.write_synthetic_data_file(tmp_path / "a.data", 17 * 12, seed=1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, x0, y0, g0, m0, u0), (_, x1, y1, g1, m1, u1) = res
    assert np.array_equal(g0, g1) and np.array_equal(m0, m1) and u0 == u1 == bytes(range(128))
    # the single-process global batch
    gds = D.load_dataset(D.get_processed_files(tmp_path), 2 * B, T, shuffle=True, seed=3)
    gx, gy = next(iter(gds))
    assert np.array_equal(gx, np.concatenate([x0, x1])) and np.array_equal(gy, np.concatenate([y0, y1]))
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), O.init_params(V, E, W, L, seed=4))
    loss, acc, G, _ = orc.loss_and_grads(gx, gy, training=False)
    ref = np.concatenate([G[k].ravel() for k in sorted(G)])
    assert np.allclose(g0, ref, atol=1e-12)
    assert abs(m0[0] - loss) < 1e-12 and abs(m0[1] - acc) < 1e-12
