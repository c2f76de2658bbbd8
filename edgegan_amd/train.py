"""``python -m edgegan_amd.train`` -- flag-compatible stand-in for ``python -m edgegan.train``
(edgegan/train.py:109-138 + the loop of edgegan/models/edgegan.py:425-489) on the HIP engine.

Under a multi-process launcher (RANK / WORLD_SIZE / LOCAL_RANK) the batches are sharded data-parallel: every rank
holds the SAME per-epoch permutation (one seed drawn by rank 0 and shared at start-up, then RandomState(seed + epoch)),
rank r trains on batch idx*world+r of it -- a partition of the epoch -- while the z draws and the gradient-penalty
alpha rows stay rank-specific; gradients are all-reduced inside the engine, rank 0 prints and saves.  The host needs
no collective library of its own: the communicator id and the seed travel through edgegan_amd.rendezvous, everything
else over RCCL."""
from __future__ import annotations

import json
import math
import os
import sys
import time

import numpy as np

from . import checkpoint, flags as F
from ._lib import Config
from .data import Dataset, Prefetcher
from .engine import EdgeGAN
from .summary import HISTOGRAM_TAGS, IMAGE_TAGS, SCALAR_TAGS, SummaryWriter


def update_flags(fl):
    """train.py:85-98."""
    if fl.input_width is None:
        fl.input_width = fl.input_height
    if fl.output_width is None:
        fl.output_width = fl.output_height
    if not fl.multiclasses:
        fl.num_classes = None
    path = os.path.join(fl.outputsroot, fl.name)
    fl.checkpoint_dir = os.path.join(path, 'checkpoints')
    fl.logdir = os.path.join(path, 'logs')
    return fl


def engine_config(fl, device: int) -> Config:
    h, w = (fl.output_height, fl.output_width) if fl.crop else (fl.input_height, fl.input_width)
    return Config(batch_size=fl.batch_size, height=h, width=w, num_classes=fl.num_classes or 0,
                  z_dim=getattr(fl, 'z_dim', 100),
                  use_image_discriminator=bool(getattr(fl, 'use_image_discriminator', True)), image_dis_size=getattr(fl, 'image_dis_size', 128),
                  use_edge_discriminator=bool(getattr(fl, 'use_edge_discriminator', True)), edge_dis_size=getattr(fl, 'edge_dis_size', 128),
                  lambda_gp=getattr(fl, 'lambda_gp', 10.0), zl_weight=getattr(fl, 'stage1_zl_loss', 10.0),
                  learning_rate=fl.learning_rate, joint_dweight=getattr(fl, 'joint_dweight', 1.0),
                  image_dweight=getattr(fl, 'image_dweight', 1.0), edge_dweight=getattr(fl, 'edge_dweight', 1.0),
                  device=device, E_norm=getattr(fl, 'E_norm', 'instance'), G_norm=getattr(fl, 'G_norm', 'instance'),
                  D_norm=getattr(fl, 'D_norm', 'instance'))


def epoch_permutation(n: int, seed: int, epoch: int) -> np.ndarray:
    """The data order of one epoch under data parallelism: identical on every rank."""
    return np.random.RandomState((seed + epoch) % (2 ** 32)).permutation(n)


def rank_batches(n_batches: int, world: int, rank: int):
    """Batch indices rank `rank` trains on: with every rank on the same permutation, a partition (drop-last)."""
    return [i * world + rank for i in range(n_batches // world)]


def main(argv=None):
    fl = update_flags(F.parse(sys.argv[1:] if argv is None else argv, F.TRAIN_FLAGS))
    F.check_supported(fl)
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', fl.gpu if world == 1 else rank))
    if 'EG_FORCE_DEVICE' in os.environ:      # test hook (as in bench.py): several ranks on one GPU
        local = int(os.environ['EG_FORCE_DEVICE'])
    if rank == 0:
        print({k: v for k, v in vars(fl).items()})
        for d in (fl.outputsroot, fl.checkpoint_dir, fl.logdir):
            os.makedirs(d, exist_ok=True)
        with open(os.path.join(fl.outputsroot, fl.name, 'flags.json'), 'w') as f:      # train.py:100-107
            json.dump({k: (v if v != math.inf else 'Infinity') for k, v in vars(fl).items()}, f, indent=4)
    dataset_config = {'input_height': fl.input_height, 'input_width': fl.input_width,
                      'output_height': fl.output_height, 'output_width': fl.output_width, 'crop': fl.crop,
                      'grayscale': False, 'z_dim': fl.z_dim}
    dataset = Dataset(fl.dataroot, fl.dataset, fl.train_size, fl.batch_size, dataset_config, fl.num_classes, 'train')
    from .variables import init_variables
    cfg = engine_config(fl, local)
    net = EdgeGAN(cfg, init_variables(cfg, seed=0))        # tf.global_variables_initializer (edgegan.py:437-440)
    shuffle_seed = None
    if world > 1:
        from . import rendezvous
        seed_bytes = rendezvous.broadcast_bytes('seed', str(np.random.randint(2 ** 31)).encode() if rank == 0 else None, rank)
        shuffle_seed = int(seed_bytes)
        rendezvous.init_data_parallel(net, rank, world)
        rendezvous.cleanup(['seed'], rank)
    writer = SummaryWriter(fl.logdir) if rank == 0 else None                      # edgegan.py:443
    counter = 1
    start_time = time.time()
    # Saver.restore raises on a missing / mis-shaped variable: so does strict=True (a silent partial restore would
    # resume on partly re-initialised weights)
    loaded, ckpt_counter = checkpoint.load(net, fl.checkpoint_dir, strict=True)   # edgegan.py:447-453
    if loaded:
        counter = ckpt_counter
        print(' [*] Load SUCCESS')
    else:
        print(' [!] Load failed...')
    for epoch in range(fl.epoch):
        if world == 1:
            dataset.shuffle()                                                        # edgegan.py:457
        else:
            if epoch == 0:
                base_order = list(dataset.data)
            dataset.data = [base_order[i] for i in epoch_permutation(len(base_order), shuffle_seed, epoch)]
        steps = len(dataset) // world
        def report(it, counter_of_it, l):
            """The host side of one iteration (edgegan.py:427-433,462-489): summaries, the printed line."""
            discriminator_err = l['joint_dis_dloss'] + l['image_dis_dloss'] + l['edge_dis_dloss']
            generator_err = l['edge_gloss'] + l['image_gloss']
            if writer is not None:                                                   # edgegan.py:427-433,462
                writer.add_scalars({t: l[t] for t in SCALAR_TAGS if t in l and (cfg.multiclasses or 'ac' not in t)},
                                   counter_of_it)
            if rank == 0:
                print('Epoch: [%2d/%2d] [%4d/%4d] time: %4.4f, joint_dis_dloss: %.8f, joint_dis_gloss: %.8f'
                      % (epoch, fl.epoch, it, steps, time.time() - start_time, 2 * discriminator_err, generator_err))

        # Default path: iteration k is ENQUEUED (pinned-memory feed, no wait), then iteration k-1 is reported -- its losses
        # were copied to the host behind it -- so decoding, printing and summary writing overlap the GPU's work and the
        # GPU never waits for the host.  The printed / logged values and their order are unchanged.  An iteration after
        # which a checkpoint is due is reported at once (the saved weights must be the ones right after that update).
        late = None                      # (it, counter) of the enqueued, not yet reported iteration
        for it, (idx, images, z, _) in enumerate(Prefetcher(dataset, rank_batches(len(dataset), world, rank))):
            if fl.reference_evaluation:      # the reference's observe-only runs: post-update weights, fresh draws, BN x6
                net.update_model(images, z)                                          # edgegan.py:461
                l = net.evaluate(bn_updates=6, tensors=writer is not None)           # edgegan.py:427-433,464-481
                if writer is not None:
                    hist = {t: 0.5 * (1.0 + np.tanh(0.5 * l[k][i].astype(np.float64))) for t, (k, i) in HISTOGRAM_TAGS.items()}   # D = sigmoid(logits)
                    hist['z'] = z
                    writer.add_histograms(hist, counter)
                    imgs = {t: l[k] for t, k in IMAGE_TAGS.items()}
                    imgs['inputs'] = images
                    writer.add_images(imgs, counter)
                report(it, counter, l)
                counter += 1
                if rank == 0 and np.mod(counter, fl.save_checkpoint_frequency) == 2:  # edgegan.py:487-489
                    checkpoint.save(net, fl.checkpoint_dir, counter)
                continue
            net.update_model_async(images, z)                                        # edgegan.py:461 (enqueue only)
            if late is not None:
                report(late[0], late[1], net.losses_async(back=1))
            late = (it, counter)
            counter += 1
            if np.mod(counter, fl.save_checkpoint_frequency) == 2:                   # edgegan.py:487-489
                report(late[0], late[1], net.losses_async(back=0))
                late = None
                if rank == 0:
                    checkpoint.save(net, fl.checkpoint_dir, counter)
        if late is not None:
            report(late[0], late[1], net.losses_async(back=0))
    if writer is not None:
        writer.close()
    net.close()


if __name__ == '__main__':
    main()
