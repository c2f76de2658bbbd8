"""One data-parallel rank of tests/test_dp_gpu.py (run as a subprocess): an engine on device 0 with its own shard of
the batch, the communicator bound through edgegan_amd.rendezvous (the product's bring-up), then either one
discriminator sub-step or whole training iterations; every variable and RMSProp slot goes to <out>/rank<r>.npz.

usage: dp_worker.py <rank> <world> <out_dir> <mode>      mode: substep1 | iterations | mismatch | order | lanes_disagree"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import edgegan_amd as ea
from edgegan_amd import rendezvous


def main():
    rank, world, out, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    cfg = ea.Config(batch_size=2)
    values = ea.init_variables(cfg, 0)                      # the same weights on every rank
    if mode == 'lanes_disagree':    # rank 1 runs another lane schedule: the bring-up must refuse (csrc/comm.cpp: comm_init)
        os.environ['EG_LANES'] = '15' if rank == 0 else '0'
    net = ea.EdgeGAN(cfg, values)
    if mode == 'lanes_disagree':
        try:
            rendezvous.init_data_parallel(net, rank, world)
        except ea.EdgeGANLibError as e:
            assert 'disagree on EG_LANES' in str(e), str(e)
            np.savez(os.path.join(out, f'rank{rank}.npz'), reported=np.ones(1))
            return
        raise SystemExit('ranks with different lane schedules were not refused')
    rendezvous.init_data_parallel(net, rank, world)
    batch = ea.synthetic_batch(cfg, seed=4321, rank=rank)   # its own shard
    net.upload_batch(**batch)
    if mode == 'mismatch':      # the ranks disagree on the collective: the stand-in must report it on every rank, not hang
        try:
            net.comm_reduce_scalars([1.0] * (2 if rank == 0 else 1))
        except ea.EdgeGANLibError as e:
            assert 'invalid usage' in str(e), str(e)
            np.savez(os.path.join(out, f'rank{rank}.npz'), reported=np.ones(1))
            return
        raise SystemExit('mismatched collective was not reported')
    if mode == 'order':    # the collectives of two iterations as this rank issued them: (group, bytes) in issue order
        net.comm_timing(True)
        net.train_step()
        net.train_step()
        net.sync()
        recs = net.comm_timing_read()
        assert all(ms >= 0.0 for _, _, ms in recs)
        np.savez(os.path.join(out, f'rank{rank}.npz'), groups=np.array([g for g, _, _ in recs]), nbytes=np.array([b for _, b, _ in recs]))
        net.close()
        return
    if mode == 'substep1':
        net.substep(1, apply=True)
    else:
        net.train_step()
        net.train_step()
    net.sync()
    net.comm_barrier()
    mx = net.comm_reduce_scalars([float(rank)], 'max')[0]   # the scalar path bench.py takes its max-over-ranks with
    assert mx == world - 1, mx
    np.savez(os.path.join(out, f'rank{rank}.npz'), **net.get_variables(slots=True))
    losses = net.losses()
    assert all(np.isfinite(v) for v in losses.values()), losses
    net.close()


if __name__ == '__main__':
    main()
