"""Thread-count sweep of the torch-CPU oracle iteration (choose bench.py's cpu_baseline settings)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for th in [int(a) for a in sys.argv[2:]] or [16, 32, 64, 128]:
    dt = bench._oracle_iteration(dict(num_classes=14), B, th)
    print(f'batch {B} threads {th}: {dt:.1f} s = {B / dt:.2f} images/s', flush=True)
