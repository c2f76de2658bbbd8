"""``python -m edgegan_amd.test`` -- flag-compatible stand-in for ``python -m edgegan.test``
(edgegan/test.py:77-126 + EdgeGAN.test, edgegan/models/edgegan.py:551-633) on the HIP engine."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import checkpoint, flags as F
from .data import Dataset, save_images
from .engine import EdgeGAN
from .train import engine_config


def update_flags(fl):
    """test.py:77-96 (batch_size is forced to 1)."""
    if fl.input_width is None:
        fl.input_width = fl.input_height
    if fl.output_width is None:
        fl.output_width = fl.output_height
    fl.batch_size = 1
    path = os.path.join(fl.outputsroot, fl.name)
    fl.checkpoint_dir = os.path.join(path, 'checkpoints')
    fl.logdir = os.path.join(path, 'logs')
    fl.test_output_dir = os.path.join(path, 'test_output')
    if not fl.multiclasses:
        fl.num_classes = None
    return fl


def pathsplit(path):
    return os.path.normpath(path).split(os.sep)


def compose(batch_images, outputL, outputR, combination, half):
    """edgegan.py:605-619 (the reference's 'outputL_inputR' branch references an undefined name)."""
    inputL = batch_images[:, :, 0:half, :]
    if combination == 'inputL_outputR':
        return np.append(inputL, outputR, axis=2)
    if combination == 'outputL_inputR':
        return np.append(outputL, batch_images[:, :, half:, :], axis=2)
    if combination == 'outputR':
        return outputR
    return np.append(np.append(batch_images, outputL, axis=2), outputR, axis=2)


def main(argv=None):
    np.random.seed(2333)                                                            # test.py:14
    fl = update_flags(F.parse(sys.argv[1:] if argv is None else argv, F.TEST_FLAGS))
    F.check_supported(fl)
    print({k: v for k, v in vars(fl).items()})
    test_root = os.path.join(fl.dataroot, fl.dataset, 'test')
    os.makedirs(os.path.join(fl.test_output_dir, fl.dataset), exist_ok=True)         # test.py:72-75
    for name in os.listdir(test_root):
        if os.path.isdir(os.path.join(test_root, name)):
            os.makedirs(os.path.join(fl.test_output_dir, fl.dataset, name), exist_ok=True)
    dataset_config = {'input_height': fl.input_height, 'input_width': fl.input_width,
                      'output_height': fl.output_height, 'output_width': fl.output_width, 'crop': fl.crop,
                      'grayscale': False}
    dataset = Dataset(fl.dataroot, fl.dataset, fl.train_size, 1, dataset_config, None, 'test')
    fl.z_dim = 100                                                                   # EdgeGAN default, edgegan.py:47
    cfg = engine_config(fl, int(fl.gpu))
    net = EdgeGAN(cfg)
    # strict: the engine starts from zero / constant weights, a checkpoint that lacks or mis-shapes a variable must fail
    # like Saver.restore (NotFoundError), not run inference on partly uninitialised networks
    loaded, _ = checkpoint.load(net, fl.checkpoint_dir, strict=True)
    if not loaded:
        print(' [!] Load failed...')                                                # edgegan.py:587-589
        return
    print(' [*] Load SUCCESS')
    half = int(fl.output_width / 2)
    for idx in range(len(dataset)):
        batch_images, filenames = dataset[idx]
        classes = None
        if fl.multiclasses:                                                          # edgegan.py:556-602
            try:
                classid = int(pathsplit(filenames[0])[-2])
            except ValueError:
                continue
            if classid >= fl.num_classes:
                continue
            classes = np.array([classid], dtype=np.int32)
        outputL, outputR = net.infer(batch_images, classes)                          # edgegan.py:606-609
        results = compose(batch_images, outputL, outputR, fl.output_combination, half)
        for fname, img in zip(filenames, results):
            parts = pathsplit(fname)
            name = os.path.join(*parts[parts.index('test') + 1:])
            save_images(img[np.newaxis, ...], os.path.join(fl.test_output_dir, fl.dataset, name))
        print('Test: [%4d/%4d]' % (idx, len(dataset)))
    net.close()


if __name__ == '__main__':
    main()
