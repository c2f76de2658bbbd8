"""Re-creates tests/golden/dataset_example from the reference checkout (data files only)."""
import os
import shutil

SRC = '/root/reference/images/dataset_example'
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dataset_example')
for phase, names in (('test', ['14809', '14810', '14811', '14812']), ('train', ['60975', '60991'])):
    os.makedirs(os.path.join(DST, phase, '3'), exist_ok=True)
    for n in names:
        shutil.copy(os.path.join(SRC, phase, n + '.png'), os.path.join(DST, phase, '3', n + '.png'))
