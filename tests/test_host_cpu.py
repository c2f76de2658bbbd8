"""Host side of the 'next' rows: flag-compatible command line (edgegan/train.py:13-75, test.py:17-65), data
pipeline (edgegan/utils/data/dataset.py, edgegan/utils/utils.py) -- no GPU needed."""
import os

import numpy as np
import pytest
from PIL import Image

from edgegan_amd import data as D
from edgegan_amd import flags as F
from edgegan_amd import test as T
from edgegan_amd import train as TR


def test_flag_syntax_matches_absl():
    fl = F.parse(['--name=exp1', '--batch_size', '8', '--nomulticlasses', '--crop=true', '--learning_rate=1e-4'],
                 F.TRAIN_FLAGS)
    assert (fl.name, fl.batch_size, fl.multiclasses, fl.crop, fl.learning_rate) == ('exp1', 8, False, True, 1e-4)
    assert fl.epoch == 100 and fl.save_checkpoint_frequency == 500 and fl.z_dim == 100 and fl.lambda_gp == 10.0
    fl = TR.update_flags(fl)
    assert fl.num_classes is None and fl.checkpoint_dir == os.path.join('outputs', 'exp1', 'checkpoints')
    t = T.update_flags(F.parse(['--name', 'x', '--output_combination=inputL_outputR', '--batch_size=64'], F.TEST_FLAGS))
    assert t.batch_size == 1 and t.test_output_dir == os.path.join('outputs', 'x', 'test_output')   # test.py:89
    with pytest.raises(SystemExit):
        F.parse(['--no_such_flag=1'], F.TRAIN_FLAGS)
    with pytest.raises(SystemExit):
        F.check_supported(F.parse(['--if_resnet_g'], F.TRAIN_FLAGS))
    cfg = TR.engine_config(F.parse(['--batch_size=4', '--nomulticlasses'], F.TRAIN_FLAGS) if False else
                           TR.update_flags(F.parse(['--batch_size=4', '--nomulticlasses'], F.TRAIN_FLAGS)), 0)
    assert (cfg.batch_size, cfg.num_classes, cfg.height, cfg.width) == (4, 0, 64, 128)
    # train.py:54-59: E_norm / G_norm / D_norm take 'instance' or 'batch' (normalization.py:12) and reach the engine's config
    fb = TR.update_flags(F.parse(['--G_norm=batch', '--D_norm', 'batch'], F.TRAIN_FLAGS))
    F.check_supported(fb)
    cb = TR.engine_config(fb, 0)
    assert (cb.E_norm, cb.G_norm, cb.D_norm) == ('instance', 'batch', 'batch')
    assert (cb.to_c().e_norm, cb.to_c().g_norm, cb.to_c().d_norm) == (0, 1, 1)
    with pytest.raises(SystemExit):
        F.check_supported(F.parse(['--E_norm=layer'], F.TRAIN_FLAGS))


def test_bytescale_is_a_min_max_stretch_for_floats():
    a = np.array([[126., 200.], [255., 126.]])
    assert D.bytescale(a).tolist() == [[0, 146], [255, 0]]          # (x-126)*255/129 + 0.5
    u8 = np.array([[3, 9]], dtype=np.uint8)
    assert D.bytescale(u8) is u8                                      # uint8 passes through untouched
    assert D.bytescale(np.full((2, 2), 7.0)).tolist() == [[0, 0], [0, 0]]


def _png(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr.astype(np.uint8)).save(path)


def test_get_image_full_range_is_identity_and_low_contrast_is_stretched(tmp_path):
    rng = np.random.RandomState(0)
    full = rng.randint(0, 256, size=(64, 128, 3))
    full[0, 0] = 0
    full[0, 1] = 255
    _png(str(tmp_path / 'a.png'), full)
    img = D.get_image(str(tmp_path / 'a.png'), 64, 128, 64, 128, crop=False)
    assert img.shape == (64, 128, 3) and np.allclose(img, full / 127.5 - 1.0)
    low = rng.randint(126, 256, size=(64, 128, 3))
    low[0, 0] = 126
    low[0, 1] = 255
    _png(str(tmp_path / 'b.png'), low)
    img = D.get_image(str(tmp_path / 'b.png'), 64, 128, 64, 128, crop=False)
    assert img.min() == -1.0 and img.max() == 1.0                   # the scipy<=1.2 imresize quirk (f3)
    with pytest.raises(ValueError):
        _png(str(tmp_path / 'g.png'), full[:, :, 0])
        D.get_image(str(tmp_path / 'g.png'), 64, 128, 64, 128, crop=False)


def _make_dataset(root, classes=(0, 1, 3), per=3):
    rng = np.random.RandomState(1)
    for c in classes:
        for i in range(per):
            _png(os.path.join(root, 'ds', 'train', str(c), f'{c}_{i}.png'), rng.randint(0, 256, size=(64, 128, 3)))
            _png(os.path.join(root, 'ds', 'test', str(c), f't{c}_{i}.png'), rng.randint(0, 256, size=(64, 128, 3)))
    _png(os.path.join(root, 'flat', 'train', 'x0.png'), rng.randint(0, 256, size=(64, 128, 3)))
    _png(os.path.join(root, 'flat', 'train', 'x1.png'), rng.randint(0, 256, size=(64, 128, 3)))


CFG = {'input_height': 64, 'input_width': 128, 'output_height': 64, 'output_width': 128, 'crop': False,
       'grayscale': False, 'z_dim': 100}


def test_dataset_contract(tmp_path):
    root = str(tmp_path)
    _make_dataset(root)
    ds = D.Dataset(root, 'ds', np.inf, 2, CFG, num_classes=4, phase='train')
    assert len(ds.data) == 9 and len(ds) == 4                        # drop-last
    np.random.seed(5)
    ds.shuffle()
    images, z, files = ds[1]
    assert images.shape == (2, 64, 128, 3) and images.dtype == np.float32 and z.shape == (2, 101)
    assert [int(z[i, -1]) for i in range(2)] == [int(os.path.basename(os.path.dirname(f))) for f in files]
    flat = D.Dataset(root, 'flat', np.inf, 2, CFG, num_classes=None, phase='train')
    assert flat[0][1].shape == (2, 100)
    test = D.Dataset(root, 'ds', np.inf, 1, CFG, None, 'test')
    assert test.data == sorted(test.data) and len(test) == 9 and len(test[0]) == 2
    with pytest.raises(Exception, match='No data found'):
        D.Dataset(root, 'missing', np.inf, 2, CFG, 4, 'train')
    with pytest.raises(Exception, match='less than the configured batch_size'):
        D.Dataset(root, 'flat', np.inf, 8, CFG, None, 'train')


def test_prefetcher_reproduces_the_synchronous_stream(tmp_path):
    root = str(tmp_path)
    _make_dataset(root)
    ds = D.Dataset(root, 'ds', np.inf, 2, CFG, num_classes=4, phase='train')
    np.random.seed(11)
    ds.shuffle()
    sync = [ds[i] for i in range(len(ds))]
    np.random.seed(11)
    ds2 = D.Dataset(root, 'ds', np.inf, 2, CFG, num_classes=4, phase='train')
    ds2.shuffle()
    got = list(D.Prefetcher(ds2, range(len(ds2))))
    assert len(got) == len(sync)
    for (idx, im, z, files), (im0, z0, files0) in zip(got, sync):
        assert files == files0 and np.array_equal(im, im0) and np.array_equal(z, z0)


def test_save_images_and_output_composition(tmp_path):
    img = np.linspace(-0.5, 0.5, 64 * 64 * 3).reshape(1, 64, 64, 3)
    D.save_images(img, str(tmp_path / 'o.png'))
    back = np.array(Image.open(str(tmp_path / 'o.png')))
    assert back.shape == (64, 64, 3) and back.min() == 0 and back.max() == 255    # imsave stretches min..max
    b = np.zeros((1, 64, 128, 3)); L = np.ones((1, 64, 64, 3)); R = 2 * np.ones((1, 64, 64, 3))
    assert T.compose(b, L, R, 'full', 64).shape == (1, 64, 256, 3)
    assert T.compose(b, L, R, 'inputL_outputR', 64).shape == (1, 64, 128, 3)
    assert T.compose(b, L, R, 'outputR', 64) is R


def test_tensorboard_event_file_round_trip(tmp_path, built_lib):
    import struct
    from edgegan_amd import summary as S
    w = S.SummaryWriter(str(tmp_path / 'logs'))
    w.add_scalars({'joint_dis_dloss': 12.5, 'zl_loss': 0.25}, 1)
    w.add_scalars({'edge_gloss': -3.0}, 2)
    w.close()
    name = os.path.basename(w.path)
    assert name.startswith('events.out.tfevents.')
    ev = S.read_events(w.path)
    assert ev[0] == (0, {}) and ev[1] == (1, {'joint_dis_dloss': 12.5, 'zl_loss': 0.25}) and ev[2] == (2, {'edge_gloss': -3.0})
    raw = open(w.path, 'rb').read()
    n0 = struct.unpack_from('<Q', raw, 0)[0]
    assert b'brain.Event:2' in raw[12:12 + n0]                      # first record = file version event
    bad = bytearray(raw)
    bad[14] ^= 0xff
    open(w.path, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        S.read_events(w.path)
    assert set(S.SCALAR_TAGS) >= {'joint_dis_dloss', 'image_dis_dloss', 'edge_dis_dloss', 'loss_d_ac', 'zl_loss'}


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_reference_example_images_through_the_pipeline():
    """The reference's own example PNGs: training pairs span 0..255 (identity through imresize), test sketches
    have min 126, so scipy<=1.2's float bytescale stretches them to the full [-1, 1] range."""
    ds = D.Dataset(GOLDEN, 'dataset_example', np.inf, 1, CFG, None, 'test')
    assert [os.path.basename(f) for f in ds.data] == ['14809.png', '14810.png', '14811.png', '14812.png']
    img, files = ds[0]
    raw = np.array(Image.open(files[0])).astype(np.float64)
    assert raw.min() == 126 and raw.max() == 255
    assert img.shape == (1, 64, 128, 3) and img.min() == -1.0 and img.max() == 1.0
    want = np.floor((raw - 126.0) * (255.0 / 129.0) + 0.5).clip(0, 255) / 127.5 - 1.0
    assert np.allclose(img[0], want, atol=1e-6)
    tr = D.Dataset(GOLDEN, 'dataset_example', np.inf, 2, CFG, 14, 'train')
    images, z, files = tr[0]
    assert images.shape == (2, 64, 128, 3) and np.all(z[:, -1] == 3)
    raw = np.array(Image.open(files[0])).astype(np.float64)
    if raw.min() == 0 and raw.max() == 255:
        assert np.allclose(images[0], raw / 127.5 - 1.0, atol=1e-6)


def test_histogram_and_image_summaries(tmp_path):
    """f4: tf.summary.histogram / tf.summary.image members of d_sum / g_sum (edgegan.py:344-411) without TF."""
    import struct
    from edgegan_amd import summary as S
    lim = S.default_bucket_limits()
    assert lim[len(lim) // 2] == 0.0 and lim[-1] == np.finfo(np.float64).max and np.all(np.diff(lim) > 0)
    assert abs(lim[len(lim) // 2 + 1] - 1e-12) < 1e-24 and abs(lim[len(lim) // 2 + 2] / lim[len(lim) // 2 + 1] - 1.1) < 1e-12
    v = np.array([-1.5, 0.0, 0.25, 0.25, 3.0])
    pb = S.histogram_proto(v)
    from edgegan_amd.tf_bundle import _pb_fields
    f = {k: x for k, _, x in _pb_fields(pb)}
    dbl = lambda x: struct.unpack('<d', struct.pack('<Q', x))[0]
    assert dbl(f[1]) == -1.5 and dbl(f[2]) == 3.0 and dbl(f[3]) == 5 and dbl(f[4]) == 2.0 and dbl(f[5]) == 11.375
    limits = np.frombuffer(f[6], dtype='<f8')
    counts = np.frombuffer(f[7], dtype='<f8')
    assert len(limits) == len(counts) and counts.sum() == 5 and np.all(np.diff(limits) > 0)
    for x in v:                                   # every value sits in the first bucket whose limit exceeds it
        i = np.searchsorted(limits, x, side='right')
        assert counts[i] >= 1
    # float image normalisation: negatives -> 128-centred, non-negative -> max to 255
    a = np.array([[[-1.0], [0.0]], [[0.5], [1.0]]], dtype=np.float32)
    assert S.normalize_float_image(a).reshape(-1).tolist() == [1, 128, 191, 255]
    assert S.normalize_float_image(np.abs(a)).reshape(-1).tolist() == [255, 0, 127, 255]
    w = S.SummaryWriter(str(tmp_path))
    w.add_histograms({'d': v, 'z': np.random.RandomState(0).normal(size=(4, 100))}, 3)
    w.add_images({'G1': np.random.RandomState(1).uniform(-1, 1, size=(5, 8, 8, 3)).astype(np.float32)}, 3)
    w.close()
    ev = S.read_event_tags(w.path)
    assert ev[1] == (3, {'d': 'histo', 'z': 'histo'})
    assert ev[2] == (3, {'G1/image/0': 'image', 'G1/image/1': 'image', 'G1/image/2': 'image'})
    S.read_events(w.path)                         # both CRCs of every record verify
