"""Where does the captured-graph iteration (EG_GRAPH=1) fail?  Runs N iterations printing progress; usage: gpu_graph_probe.py <classes> <batch> <n> <sync 0/1>"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import edgegan_amd as ea
classes, batch, n, sync = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = ea.Config(batch_size=batch, num_classes=classes, compute_dtype=sys.argv[5] if len(sys.argv) > 5 else 'f32x3')
net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
net.upload_batch(**ea.synthetic_batch(cfg))
for i in range(n):
    net.train_step()
    if sync:
        net.sync()
    print('step', i, net.graph_info(), flush=True)
net.sync()
print('losses', net.losses(), flush=True)
net.close()
print('closed', flush=True)
