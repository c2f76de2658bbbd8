"""Data-parallel contract on CPU (gloo, world_size 2): each rank computes the gradient of its own shard,
the flat group gradient is all-reduced (sum) and scaled by 1/nranks before RMSProp -- the protocol
edgegan_amd/csrc/model.cpp:apply_group runs over RCCL.  Checked against a single-process computation
and for replica identity."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import edgegan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model_and_shards():
    cfg = O.Config(batch_size=1, num_classes=0, gf_dim=8, df_dim=8)
    rng = np.random.RandomState(7)
    vals = {}
    for s in O.variable_specs(cfg):
        kind, _, par = s.init.partition(':')
        vals[s.name] = np.full(s.shape, float(par)) if kind == 'const' else rng.normal(size=s.shape) * 0.05
    shards = []
    for r in range(2):
        shards.append(O.StepInputs(torch.tensor(rng.uniform(-1, 1, size=(1, 64, 128, 3))),
                                   torch.tensor(rng.normal(size=(1, 100))), torch.tensor(rng.uniform(size=(3, 1))),
                                   torch.tensor([0.25])))
    return cfg, vals, shards


def _flat(grads, names):
    return torch.cat([grads[n].reshape(-1) for n in names])


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg, vals, shards = _model_and_shards()
    m = O.OracleModel(cfg, vals, dtype=torch.float64)
    grads, _ = m.compute_grads('d_optim', shards[rank], shards[rank].eps[0])
    names = sorted(grads)
    flat = _flat(grads, names)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)      # one collective per optimised group
    flat /= world
    off = 0
    for n in names:
        k = grads[n].numel()
        grads[n] = flat[off:off + k].reshape(grads[n].shape)
        off += k
    m.apply_grads(grads)
    torch.save({n: m.p[n] for n in names}, os.path.join(out, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_average_then_rmsprop_keeps_replicas_identical(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / 'rank0.pt')
    b = torch.load(tmp_path / 'rank1.pt')
    for n in a:
        assert torch.equal(a[n], b[n]), n                     # replicas bit-identical after the update
    # single-process reference: average of the two shard gradients, same RMSProp
    cfg, vals, shards = _model_and_shards()
    m = O.OracleModel(cfg, vals, dtype=torch.float64)
    g0, _ = m.compute_grads('d_optim', shards[0], shards[0].eps[0])
    g1, _ = m.compute_grads('d_optim', shards[1], shards[1].eps[0])
    m.apply_grads({n: (g0[n] + g1[n]) / 2 for n in g0})
    for n in a:
        assert torch.allclose(a[n], m.p[n], rtol=0, atol=1e-15), n


def test_shard_means_equal_global_batch_gradient():
    """Every loss is a batch mean and IN / GP norms are per sample: the mean of per-shard
    gradients IS the global-batch gradient for the discriminator step (the generator's BatchNorm is the one
    cross-sample op, so feed fixed fakes)."""
    cfg = O.Config(batch_size=2, num_classes=0, df_dim=8)
    rng = np.random.RandomState(3)
    p = {s.name: torch.tensor(rng.normal(size=s.shape) * 0.1, requires_grad=True)
         for s in O.discriminator_specs('D', cfg, 64, 128)}
    real, fake = torch.tensor(rng.uniform(-1, 1, size=(2, 64, 128, 3))), torch.tensor(rng.uniform(-1, 1, size=(2, 64, 128, 3)))
    alpha = torch.tensor(rng.uniform(size=(2,)))

    def loss(sl):
        d_r = O.discriminator(p, 'D', real[sl])[1]
        d_f = O.discriminator(p, 'D', fake[sl])[1]
        return O.discriminator_ganloss(d_f, d_r) + O.gradient_penalty_term(p, 'D', fake[sl], real[sl], alpha[sl], 10.0)
    names = sorted(p)
    g_all = torch.autograd.grad(loss(slice(0, 2)), [p[n] for n in names])
    g_a = torch.autograd.grad(loss(slice(0, 1)), [p[n] for n in names])
    g_b = torch.autograd.grad(loss(slice(1, 2)), [p[n] for n in names])
    for n, ga, gb, gg in zip(names, g_a, g_b, g_all):
        assert torch.allclose((ga + gb) / 2, gg, rtol=1e-9, atol=1e-12), n


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment starts the two ranks itself
    (bench.py:self_launch; the GPU leg is tests/test_dp_gpu.py::test_bench_launches_its_own_ranks).  Without a GPU both
    ranks must fail LOUDLY in eg_create -- the product has no CPU fallback -- and the launcher must pass that on as a
    non-zero status with no JSON line, instead of hanging in a rendezvous."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU box: covered by tests/test_dp_gpu.py')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert r.returncode != 0
    assert 'no CPU fallback' in r.stderr and 'stopping the other ranks' in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def _shard_worker(rank, world, key, root, out):
    """What edgegan_amd/train.py does per rank before and inside an epoch, minus the GPU: rendezvous of the shuffle
    seed (file-based, no torch), the shared permutation, this rank's batches."""
    os.environ['EG_RDZV_KEY'] = key
    os.environ['EG_RDZV_DIR'] = out
    sys.path.insert(0, ROOT)
    from edgegan_amd import rendezvous
    from edgegan_amd.data import Dataset
    from edgegan_amd.train import epoch_permutation, rank_batches
    np.random.seed(1000 + rank)          # the ranks' global np.random streams differ (they are unseeded in production)
    seed = int(rendezvous.broadcast_bytes('seed', str(np.random.randint(2 ** 31)).encode() if rank == 0 else None, rank))
    dcfg = {'input_height': 64, 'input_width': 128, 'output_height': 64, 'output_width': 128, 'crop': False, 'z_dim': 100}
    ds = Dataset(root, 'toy', np.inf, 2, dcfg, 2, 'train')
    base = list(ds.data)
    seen = []
    for epoch in range(2):
        ds.data = [base[i] for i in epoch_permutation(len(base), seed, epoch)]
        files = [f for b in rank_batches(len(ds), world, rank) for f in ds.filenames(b)]
        z = ds.draw_z(ds.filenames(0))
        seen.append({'files': files, 'z0': z[:, :3].tolist(), 'order': list(ds.data)})
    import json
    with open(os.path.join(out, f'shard{rank}.json'), 'w') as f:
        json.dump({'seed': seed, 'epochs': seen}, f)


def test_data_parallel_shards_are_a_partition_of_every_epoch(tmp_path):
    """with an unseeded per-rank shuffle the shards overlapped.  Now: one seed from rank 0, identical
    per-epoch permutation on every rank, rank r takes batches i*world + r => disjoint, jointly covering the epoch
    (drop-last), different order per epoch; the z draws stay rank-specific."""
    import json
    root = tmp_path / 'data'
    for c in (0, 1):
        os.makedirs(root / 'toy' / 'train' / str(c))
        for i in range(9):
            (root / 'toy' / 'train' / str(c) / f'{i}.png').write_bytes(b'')     # names only: nothing is decoded here
    out = str(tmp_path)
    mp.spawn(_shard_worker, args=(2, f'test{os.getpid()}', str(root), out), nprocs=2, join=True)
    r = [json.load(open(os.path.join(out, f'shard{k}.json'))) for k in (0, 1)]
    assert r[0]['seed'] == r[1]['seed']
    for e in (0, 1):
        a, b = r[0]['epochs'][e], r[1]['epochs'][e]
        assert a['order'] == b['order']                               # same permutation on both ranks
        assert not set(a['files']) & set(b['files'])                  # disjoint shards
        assert len(a['files']) == len(b['files']) == 8                # 18 files -> 9 batches of 2 -> 4 per rank
        assert set(a['files']) | set(b['files']) <= set(a['order']) and len(set(a['files']) | set(b['files'])) == 16
        assert a['z0'] != b['z0']                                     # z stays rank-specific
    assert r[0]['epochs'][0]['order'] != r[0]['epochs'][1]['order']   # reshuffled per epoch
