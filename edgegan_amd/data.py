"""Host-side data path of the reference (edgegan/utils/data/dataset.py:18-89, edgegan/utils/utils.py:41-54,
126-164) with an asynchronous prefetcher in front of the GPU (the reference loads synchronously).

The reference reads images through scipy<=1.2's ``scipy.misc`` helpers, whose float handling is load-bearing:
``imresize`` / ``imsave`` first run ``bytescale`` on a float array -- a MIN-MAX CONTRAST STRETCH to 0..255 --
(SURVEY.md 8f row f3).  Those helpers are restated here on PIL."""
from __future__ import annotations

import os
import queue
from glob import glob
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
from PIL import Image


# ---- scipy.misc (<=1.2) semantics -------------------------------------------------------------------
def bytescale(data: np.ndarray, cmin=None, cmax=None, high=255, low=0) -> np.ndarray:
    """scipy/misc/pilutil.py: uint8 passes through; anything else is stretched from [min,max] to [low,high]."""
    if data.dtype == np.uint8:
        return data
    cmin = data.min() if cmin is None else cmin
    cmax = data.max() if cmax is None else cmax
    cscale = cmax - cmin
    if cscale < 0:
        raise ValueError('`cmax` should be larger than `cmin`.')
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def imread(path: str) -> np.ndarray:
    """scipy.misc.imread(path).astype(np.float)  (utils.py:126-130)."""
    im = Image.open(path)
    if im.mode == 'P':
        im = im.convert('RGBA' if 'transparency' in im.info else 'RGB')
    elif im.mode == '1':
        im = im.convert('L')
    return np.array(im).astype(np.float64)


def imresize(arr: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """scipy.misc.imresize(arr, [h, w]) with the default interp='bilinear': toimage (bytescale) -> PIL resize."""
    data = bytescale(np.asarray(arr))
    im = Image.fromarray(data)
    im = im.resize((size[1], size[0]), resample=Image.BILINEAR)
    return np.array(im)


def transform(image: np.ndarray, input_height, input_width, resize_height, resize_width, crop) -> np.ndarray:
    """utils.py:147-160."""
    if crop:
        h, w = image.shape[:2]
        j = int(round((h - input_height) / 2.))
        i = int(round((w - input_width) / 2.))
        out = imresize(image[j:j + input_height, i:i + input_width], [resize_height, resize_width])
    else:
        out = imresize(image, [resize_height, resize_width])
    return np.array(out) / 127.5 - 1.


def get_image(path, input_height, input_width, resize_height=64, resize_width=64, crop=True) -> np.ndarray:
    """utils.py:41-50."""
    img = transform(imread(path), input_height, input_width, resize_height, resize_width, crop)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f'{path}: expected an RGB image, got array of shape {img.shape}')
    return img


def save_images(images: np.ndarray, path: str):
    """utils.py:53-54,133-135 with size [1,1]: inverse_transform -> merge -> squeeze -> scipy.misc.imsave, whose
    toimage() stretches the float image's min..max to 0..255 again."""
    img = np.squeeze((np.asarray(images) + 1.) / 2.)
    Image.fromarray(bytescale(img)).save(path)


# ---- dataset ----------------------------------------------------------------------------------------
class Dataset:
    """dataset.py:18-89.  train: <dataroot>/<name>/train/<class>/*.{png,jpg} (or .../train/*.png without
    classes), glob order then np.random.shuffle per epoch, drop-last batches, z ~ N(0,1) with the class id (the
    directory name) appended; test: every png/jpg under <dataroot>/<name>/test, sorted."""

    def __init__(self, dataroot, name, size, batchsize, config: Dict, num_classes=None, phase='train'):
        assert phase in ('train', 'test')
        self.batchsize, self.num_classes, self.config, self.phase = batchsize, num_classes, config, phase
        if phase == 'train':
            if num_classes is not None:
                self.data: List[str] = []
                for i in range(num_classes):
                    for ext in ('*.png', '*.jpg'):
                        self.data.extend(glob(os.path.join(dataroot, name, phase, str(i), ext)))
                data_path = os.path.join(dataroot, name, phase)
            else:
                data_path = os.path.join(dataroot, name, phase, '*.png')
                self.data = glob(data_path)
        else:
            data_path = os.path.join(dataroot, name, phase)
            self.data = []
            for ext in ('*.png', '*.jpg'):
                self.data.extend(str(p) for p in Path(data_path).rglob(ext))
            self.data = sorted(self.data)
        if len(self.data) == 0:
            raise Exception("[!] No data found in '" + data_path + "'")
        if len(self.data) < self.batchsize:
            raise Exception('[!] Entire dataset size is less than the configured batch_size')
        self.size = int(min(len(self.data), size))

    def shuffle(self):
        np.random.shuffle(self.data)

    def __len__(self):
        return self.size // self.batchsize

    def filenames(self, idx) -> List[str]:
        return self.data[idx * self.batchsize:(idx + 1) * self.batchsize]

    def load_images(self, filenames: List[str]) -> np.ndarray:
        c = self.config
        batch = [get_image(f, c['input_height'], c['input_width'], c['output_height'], c['output_width'], c['crop'])
                 for f in filenames]
        return np.array(batch).astype(np.float32)

    def draw_z(self, filenames: List[str]) -> np.ndarray:
        z = np.random.normal(size=(self.batchsize, self.config['z_dim']))
        if self.num_classes is not None:
            def get_class(p):
                end = p.rfind('/')
                return int(p[p.rfind('/', 0, end) + 1:end])
            cls = np.array([get_class(f) for f in filenames]).reshape((self.batchsize, 1))
            z = np.concatenate((z, cls), axis=1)
        return z

    def __getitem__(self, idx):
        files = self.filenames(idx)
        images = self.load_images(files)
        if self.phase == 'train':
            return images, self.draw_z(files), files
        assert images.shape[0] == len(files)
        return images, files


class Prefetcher:
    """Decodes the next batches on worker threads while the GPU trains on the current one: `workers` threads each decode
    whole batches (PIL releases the GIL inside the PNG / JPEG decoders and the resize), results are handed out strictly in
    batch order, at most `depth` batches ahead.  z is drawn on the consumer side in batch order, so the np.random stream
    is the reference's (shuffle, then one draw per batch)."""

    def __init__(self, dataset: Dataset, indices: List[int], depth: int = 6, workers: Optional[int] = None):
        self.ds, self.indices = dataset, list(indices)
        self.workers = workers if workers is not None else max(1, min(8, (os.cpu_count() or 2) - 1))
        self.depth = max(depth, self.workers)

    def _load(self, idx):
        files = self.ds.filenames(idx)
        return idx, self.ds.load_images(files), files

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        pending: 'queue.deque' = __import__('collections').deque()
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix='edgegan-decode') as pool:
            it = iter(self.indices)
            for idx in it:
                pending.append(pool.submit(self._load, idx))
                if len(pending) >= self.depth:
                    break
            while pending:
                idx, images, files = pending.popleft().result()      # decoding errors surface here, in the training thread
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(self._load, nxt))
                yield idx, images, self.ds.draw_z(files), files
