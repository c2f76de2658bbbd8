#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -5
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
python - <<'PY'
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import edgegan_amd as ea
for env in ('1', ''):
    if env: os.environ['EG_NO_CONV_C3'] = env
    else: os.environ.pop('EG_NO_CONV_C3', None)
PY
for d in 1 0; do
  if [ $d = 1 ]; then export EG_NO_CONV_C3=1; else unset EG_NO_CONV_C3; fi
  python - <<'PY'
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import edgegan_amd as ea
net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype='f32x3'))
rng = np.random.RandomState(0)
for (NB, H, N) in [(64, 32, 128), (64, 16, 256), (64, 8, 512), (64, 64, 128), (64, 8, 768)]:
    x = rng.normal(size=(NB, H, H, 3)).astype(np.float32)
    w = (rng.normal(size=(3, 3, 3, N)) * 0.05).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    args = ('conv2d', [x, w, b, None], [NB, H, H, 3, N, 3, 1, 0], [(NB, H, H, N), None, None, None])
    net.test_op(*args)
    net.profile_begin()
    for _ in range(3): y = net.test_op(*args)[0]
    k = net.profile_end()['kernels']
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = sum(np.einsum('bhwc,cn->bhwn', xp[:, i:i + H, j:j + H], w[i, j].astype(np.float64)) for i in range(3) for j in range(3)) + b
    print('EG_NO_CONV_C3=%s' % os.environ.get('EG_NO_CONV_C3', '0'), (NB, H, N), [(q['name'], round(q['ms'] / q['launches'] * 1e3, 1)) for q in k], 'err %.2e' % (np.abs(y - ref).max() / np.abs(ref).max()))
net.close()
PY
done 2>&1 | tee gpurun_out/r03x/conv_c3_micro.txt
