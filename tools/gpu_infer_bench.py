"""configs[1]: single-class 64x128 generator-only inference (edgegan.test path: E -> G1, E -> G2), batch 64.
The C ABI takes and returns HOST buffers (eg_infer), so the rate printed here is PCIe-inclusive."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea

for dtype in ('f32', 'bf16'):
    cfg = ea.Config(batch_size=64, num_classes=0, compute_dtype=dtype)
    net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
    imgs = ea.synthetic_batch(cfg, seed=1)['images']
    for _ in range(3):
        net.infer(imgs, None, eps2=[0.1, 0.2])
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        net.infer(imgs, None, eps2=[0.1, 0.2])
    dt = time.perf_counter() - t0
    print(f'{dtype}: {64 * n / dt:9.1f} images/s  ({dt / n * 1e3:.2f} ms per batch of 64, host buffers in and out)')
    net.close()
