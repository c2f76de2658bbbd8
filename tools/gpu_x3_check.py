"""fp32-via-three-bf16-limbs mode (compute_dtype='f32x3') against the exact-fp32 MFMA mode: error of every conv / deconv
op case (forward, input gradient, filter gradient) relative to the fp64 oracle, side by side."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import edgegan_amd as ea
import parity


def rows_for(net):
    rows = []
    for c in parity.CONV_CASES:
        rows += parity.check_conv(net, c)
    for c in parity.DECONV_CASES:
        rows += parity.check_deconv(net, c)
    return rows + parity.real_shape_checks(net) + parity.long_m_checks(net)


def main():
    res = {}
    for mode in ('f32', 'f32x3'):
        net = ea.EdgeGAN(ea.Config(batch_size=2, compute_dtype=mode))
        res[mode] = rows_for(net)
        net.close()
    worst = {m: 0.0 for m in res}
    print(f'{"case":70s} {"f32 err":>10s} {"f32x3 err":>10s} {"tol":>8s}')
    for a, b in zip(res['f32'], res['f32x3']):
        assert a[0] == b[0]
        flag = '' if b[1] <= b[2] else '  FAIL'
        print(f'{a[0]:70s} {a[1]:10.2e} {b[1]:10.2e} {a[2]:8.0e}{flag}')
        worst['f32'] = max(worst['f32'], a[1] / a[2])
        worst['f32x3'] = max(worst['f32x3'], b[1] / b[2])
    print('worst error / tolerance:', worst)


if __name__ == '__main__':
    main()
