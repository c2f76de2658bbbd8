"""Direct-definition (numpy loops) restatements of the TF-1.14 op semantics the oracle relies on --
TEST INFRASTRUCTURE.  These follow the published TF algorithms literally and
exist only to pin oracle/edgegan_oracle.py's torch formulations on small cases; they are the known-answer
layer of a parity chain that the reference itself cannot pin (PARITY UNPINNED: no TF, no goldens).
"""
import math

import numpy as np


def same_pad(n, k, s):
    """tensorflow/core/framework/common_shape_fns.cc GetWindowedOutputSize (SAME)."""
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def conv2d_same(x, w, s):
    """tf.nn.conv2d(x NHWC, w HWIO, strides s, 'SAME') by definition."""
    N, H, W, C = x.shape
    k, _, _, Co = w.shape
    OH, pt = same_pad(H, k, s)
    OW, pl = same_pad(W, k, s)
    y = np.zeros((N, OH, OW, Co))
    for oy in range(OH):
        for ox in range(OW):
            for kh in range(k):
                for kw in range(k):
                    iy, ix = oy * s + kh - pt, ox * s + kw - pl
                    if 0 <= iy < H and 0 <= ix < W:
                        y[:, oy, ox, :] += x[:, iy, ix, :] @ w[kh, kw]
    return y


def conv2d_reflect(x, w):
    """tf.pad(x, p, 'REFLECT') + VALID conv (conv.py:23-26); REFLECT excludes the border pixel."""
    k = w.shape[0]
    p = (k - 1) // 2
    N, H, W, C = x.shape

    def refl(i, n):
        i = -i if i < 0 else i
        return 2 * (n - 1) - i if i >= n else i
    y = np.zeros((N, H, W, w.shape[3]))
    for oy in range(H):
        for ox in range(W):
            for kh in range(k):
                for kw in range(k):
                    y[:, oy, ox, :] += x[:, refl(oy + kh - p, H), refl(ox + kw - p, W), :] @ w[kh, kw]
    return y


def conv2d_transpose_same_k5s2(x, w):
    """tf.nn.conv2d_transpose(x, w[kh,kw,Cout,Cin], out=2*in, strides 2, SAME) = the input gradient of
    the forward k5-s2-SAME conv whose pad_before is 1: y[2i+k-1] += x[i] w[k]."""
    N, H, W, Ci = x.shape
    Co = w.shape[2]
    y = np.zeros((N, 2 * H, 2 * W, Co))
    for i in range(H):
        for j in range(W):
            for kh in range(5):
                for kw in range(5):
                    oy, ox = 2 * i + kh - 1, 2 * j + kw - 1
                    if 0 <= oy < 2 * H and 0 <= ox < 2 * W:
                        y[:, oy, ox, :] += x[:, i, j, :] @ w[kh, kw].T
    return y


def resize_bicubic_legacy(x, size):
    """core/kernels/resize_bicubic_op.cc, align_corners=False, half_pixel_centers=False: per output pixel,
    4x4 taps from a 1024-entry float table (A=-0.75), indices clamped to the image."""
    a = -0.75
    tab = np.zeros((1025, 2), dtype=np.float32)
    for i in range(1025):
        t = i / 1024.0
        tab[i, 0] = ((a + 2) * t - (a + 3)) * t * t + 1
        t += 1.0
        tab[i, 1] = ((a * t - 5 * a) * t + 8 * a) * t - 4 * a
    N, H, W, C = x.shape

    def taps(o, n_in, n_out):
        loc = o * (n_in / n_out)
        fl = math.floor(loc)
        off = int(np.rint((loc - fl) * 1024))
        wts = [tab[off, 1], tab[off, 0], tab[1024 - off, 0], tab[1024 - off, 1]]
        idx = [min(max(fl - 1 + t, 0), n_in - 1) for t in range(4)]
        return idx, wts
    y = np.zeros((N, size, size, C))
    for oy in range(size):
        iy, wy = taps(oy, H, size)
        for ox in range(size):
            ix, wx = taps(ox, W, size)
            for a_ in range(4):
                for b_ in range(4):
                    y[:, oy, ox, :] += float(wy[a_]) * float(wx[b_]) * x[:, iy[a_], ix[b_], :]
    return y


def avg_pool_same(x, k):
    """tf.nn.avg_pool(k, stride k, SAME): mean over the VALID elements of each window."""
    N, H, W, C = x.shape
    OH, pt = same_pad(H, k, k)
    OW, pl = same_pad(W, k, k)
    y = np.zeros((N, OH, OW, C))
    for oy in range(OH):
        for ox in range(OW):
            ys = [i for i in range(oy * k - pt, oy * k - pt + k) if 0 <= i < H]
            xs = [j for j in range(ox * k - pl, ox * k - pl + k) if 0 <= j < W]
            y[:, oy, ox, :] = x[:, ys][:, :, xs].mean(axis=(1, 2))
    return y


def rmsprop(w, g, ms, mom, lr, decay=0.9, momentum=0.0, eps=1e-10):
    """core/kernels/training_ops.cc ApplyRMSProp."""
    ms = ms + (g * g - ms) * (1 - decay)
    mom = mom * momentum + lr * g / np.sqrt(ms + eps)
    return w - mom, ms, mom
