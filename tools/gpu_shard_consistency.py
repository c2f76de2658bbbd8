import sys, os
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import numpy as np, edgegan_amd as ea, parity
names = ['D2/Conv/weights', 'D2/mru_conv_unit_t_1_layer_0/Conv_2/weights', 'D2/mru_conv_unit_t_1_layer_0/update_gate/weights',
         'D2/mru_conv_unit_t_2_layer_0/Conv_1/weights', 'D2/mru_conv_unit_t_4_layer_0/Conv_2/weights',
         'D2/mru_conv_unit_t_3_layer_0/Conv_3/biases', 'D2/mru_conv_unit_t_2_layer_0/norm_activation_in/prelu/param',
         'D2/fully_connected/weights']
big_cfg, small_cfg = ea.Config(batch_size=64), ea.Config(batch_size=2)
values = ea.init_variables(big_cfg, 0)
batch = ea.synthetic_batch(big_cfg)
big = ea.EdgeGAN(big_cfg, values); big.upload_batch(**batch); big.substep(4, apply=False)
want = {n: big.get_variable(n + '/grad').astype(np.float64) for n in names}; big.close()
small = ea.EdgeGAN(small_cfg, values)
acc = {n: 0.0 for n in names}
for s in range(32):
    sl = slice(2 * s, 2 * s + 2)
    small.upload_batch(batch['images'][sl], batch['z'][sl], batch['alpha'][:, sl].copy(), batch['eps'])
    small.substep(4, apply=False)
    for n in names: acc[n] = acc[n] + small.get_variable(n + '/grad').astype(np.float64) / 32
for n in names: print(f'{parity.rel_l2(want[n], acc[n]):.3e}  {n}')
