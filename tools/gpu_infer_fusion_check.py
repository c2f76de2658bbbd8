"""Forward-only fusions of round 6 (residual tail, encoder head, tanh in the last transposed convolution's store, BatchNorm's
plane store) against the SAME graph with each of them switched off (experiments build: EG_NO_RES_TAIL,
EG_NO_ENC_HEAD, EG_NO_EPI_TANH): the element-wise fusions must not change a bit; the encoder head re-orders a 512-term fp32
sum of exact products, so its outputs are compared in max-norm.  Runs the three compute modes at batch 8 and 64, single- and 14-class.
usage (GPU box, after `make -C edgegan_amd/csrc EXPERIMENTS=1`): python tools/gpu_infer_fusion_check.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, 'edgegan_amd', 'libedgegan_hip_exp.so')


def worker():
    sys.path.insert(0, ROOT)
    import numpy as np
    import edgegan_amd as ea
    out = {}
    for dtype in ('f32', 'f32x3', 'bf16'):
        for classes, nb in ((0, 8), (14, 8), (0, 64)):
            cfg = ea.Config(batch_size=nb, num_classes=classes, compute_dtype=dtype)
            net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 3))
            imgs = np.random.RandomState(nb + classes).uniform(-1, 1, size=(nb, 64, 128, 3)).astype(np.float32)
            cls = (np.arange(nb) % max(classes, 1)).astype(np.int32)
            e, i = net.infer(imgs, cls, eps2=[0.3, -0.7])
            out[f'{dtype}_c{classes}_b{nb}_e'] = e
            out[f'{dtype}_c{classes}_b{nb}_i'] = i
            net.close()
    np.savez(sys.argv[2], **out)


def main():
    import numpy as np
    runs = {'shipped': {}, 'exp_all_on': {'EG_LIB_PATH': EXP},
            'no_res_tail': {'EG_LIB_PATH': EXP, 'EG_NO_RES_TAIL': '1'}, 'no_epi_tanh': {'EG_LIB_PATH': EXP, 'EG_NO_EPI_TANH': '1'},
            'no_enc_head': {'EG_LIB_PATH': EXP, 'EG_NO_ENC_HEAD': '1'},
            'all_off': {'EG_LIB_PATH': EXP, 'EG_NO_RES_TAIL': '1', 'EG_NO_EPI_TANH': '1', 'EG_NO_ENC_HEAD': '1'}}
    res = {}
    for name, env in runs.items():
        path = f'/tmp/fusion_{name}.npz'
        subprocess.run([sys.executable, os.path.abspath(__file__), 'worker', path], env=dict(os.environ, **env), check=True)
        res[name] = dict(np.load(path))
    ref = res['shipped']
    ok = True
    for name in ('exp_all_on', 'no_res_tail', 'no_epi_tanh'):
        same = all(np.array_equal(ref[k], res[name][k]) for k in ref)
        ok &= same
        print(f'{name:14s} vs shipped: {"bit-identical" if same else "DIFFERS"} over {len(ref)} outputs')
    for name in ('no_enc_head', 'all_off'):
        worst = max(float(np.abs(ref[k] - res[name][k]).max()) for k in ref)
        by_mode = {m: max(float(np.abs(ref[k] - res[name][k]).max()) for k in ref if k.startswith(m + '_')) for m in ('f32', 'f32x3', 'bf16')}
        # fp32 modes: a re-ordered 512-term sum (1e-6); bf16 mode: that 1e-7-relative change of z moves ~1e-4 of the generator's
        # operands across a bf16 rounding boundary (2^-8 relative each) -- the mode's own noise floor, as in tests/test_bf16_gpu.py
        ok &= by_mode['f32'] < 2e-5 and by_mode['f32x3'] < 2e-5 and by_mode['bf16'] < 1e-2
        print(f'{name:14s} vs shipped: max |difference| of the tanh outputs {worst:.2e} (by mode: {by_mode})')
    print('OK' if ok else 'FAILED')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    worker() if len(sys.argv) > 1 and sys.argv[1] == 'worker' else main()
