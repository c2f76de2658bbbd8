"""The two analysis tools the kernels' correctness / measurement arguments lean on, exercised on CPU:

  * tools/isa_async_lint.py -- the ISA check for inline-assembly LDS reads whose destination is read or re-used before the
    wait that retires the read (the root cause of round 3's S2D one-plane split-K bug): a planted instance of each hazard must
    be found, the corrected sequence must pass, and -- when hipcc is there -- the shipped kernels must lint clean;
  * tools/trace_overlap.py -- on a synthetic rocpd trace with known overlap."""
import importlib.util
import os
import shutil
import sqlite3
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _kernel(body):
    return [(i, l) for i, l in enumerate(body.strip('\n').split('\n'), 1)]


CLOBBER = '''
	;;#ASMSTART
	ds_read_b128 v[10:13], v2 offset:0
	;;#ASMEND
	v_add_u32_e32 v10, v4, v5
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_endpgm
'''
READ_EARLY = '''
	;;#ASMSTART
	ds_read_b64_tr_b16 v[20:21], v3 offset:64
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[30:33], a[0:15]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_endpgm
'''
CLEAN = '''
	;;#ASMSTART
	ds_read_b128 v[10:13], v2 offset:0
	;;#ASMEND
	;;#ASMSTART
	ds_read_b128 v[14:17], v2 offset:2048
	;;#ASMEND
	v_add_u32_e32 v6, v4, v5
	;;#ASMSTART
	s_waitcnt lgkmcnt(1)
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[30:33], a[0:15]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 a[0:15], v[14:17], v[30:33], a[0:15]
	s_endpgm
'''
PARTIAL_WAIT = '''
	;;#ASMSTART
	ds_read_b128 v[10:13], v2 offset:0
	;;#ASMEND
	;;#ASMSTART
	ds_read_b128 v[14:17], v2 offset:2048
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt lgkmcnt(1)
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 a[0:15], v[14:17], v[30:33], a[0:15]
	s_endpgm
'''
LOOP_CARRIED = '''
.LBB0_1:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[30:33], a[0:15]
	;;#ASMSTART
	ds_read_b128 v[10:13], v2 offset:0
	;;#ASMEND
	v_mov_b32_e32 v12, v7
	s_cbranch_scc1 .LBB0_1
	s_endpgm
'''


def test_async_lint_finds_planted_hazards_and_passes_the_corrected_sequences():
    L = _load('isa_async_lint')
    kinds = lambda body: sorted({f[1] for f in L.lint_kernel('k', _kernel(body))})
    assert kinds(CLOBBER) == ['clobber-before-land']          # the round-3 bug: a dead destination re-used as a scratch register
    assert kinds(READ_EARLY) == ['read-before-land']
    assert kinds(CLEAN) == []                                  # in-order returns: lgkmcnt(1) retires the OLDER of two reads
    assert kinds(PARTIAL_WAIT) == ['read-before-land']         # ... and not the newer one
    assert kinds(LOOP_CARRIED) == ['clobber-before-land']      # found through the back edge's state, within the loop body


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
def test_shipped_kernels_with_inline_assembly_reads_lint_clean(tmp_path):
    """The kernels whose fragment reads are inline assembly: the new packed filter-gradient kernel (compiles in seconds; the
    halo kernels take minutes and are linted by hand when they change: DESIGN.md 3.10)."""
    src = os.path.join(ROOT, 'edgegan_amd', 'csrc', 'k_thin_wgrad.hip')
    out = tmp_path / 'k.s'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', src, '-o', str(out)],
                   check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_async_lint.py'), str(out)], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if 'finding(s)' in l]
    assert len(lines) >= 8 and all(l.endswith(': 0 finding(s)') for l in lines), r.stdout[-2000:]
    assert r.returncode == 0


def test_trace_overlap_on_a_synthetic_trace(tmp_path, capsys):
    def make(path, rows):
        con = sqlite3.connect(path)
        con.execute('create table kernels (name text, start integer, end integer, grid_x integer, grid_y integer, grid_z integer, '
                    'workgroup_x integer, workgroup_y integer, workgroup_z integer, lds_size integer, vgpr_count integer, accum_vgpr_count integer)')
        con.executemany('insert into kernels values (?,?,?,?,?,?,?,?,?,?,?,?)', rows)
        con.commit()
        con.close()
    big = (512 * 256, 1, 1, 512, 1, 1, 150 * 1024, 200, 0)      # 256 blocks of a one-block-per-CU kernel: fills the chip
    small = (64 * 16, 1, 1, 64, 1, 1, 0, 32, 0)                  # 16 one-wave blocks: ~ nothing
    # concurrent run: A [0, 100), B [50, 150) overlap by 50; C [200, 220) alone after a gap of 50
    make(str(tmp_path / 'c.db'), [('A', 0, 100) + big, ('B', 50, 150) + big, ('C', 200, 220) + small])
    make(str(tmp_path / 's.db'), [('A', 0, 80) + big, ('B', 80, 130) + big, ('C', 130, 150) + small])
    T = _load('trace_overlap')
    argv = sys.argv
    sys.argv = ['trace_overlap.py', str(tmp_path / 'c.db'), str(tmp_path / 's.db'), '5']
    try:
        T.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert '>= 1 kernel running 0.00 ms (77.3 %)' in out           # 170 of 220 ns
    assert "{0: '22.7 %', 1: '54.5 %', 2: '22.7 %'}" in out
    assert 'block-slot fill' in out and '68.' in out                # 150 ns full + 20 ns x 16 / (256 x 8 x 4) of 220
    a = [l for l in out.splitlines() if l.startswith('A ')][0].split()
    assert abs(float(a[-1]) - 1.25) < 1e-6                          # A: 100 concurrent vs 80 serialised


def test_bench_roofline_object_and_side_leg_reduction(tmp_path, monkeypatch):
    """bench.py's pure host logic (no GPU): the `roofline` object built from the engine's per-kernel event records -- dominant
    kernel = the bucket with the most time, limb kernels priced at 6 executed bf16 MFMA flops per fp32 flop against the dense bf16
    peak -- and a side leg's reduction of a child run's JSON line (a failing child must never hide the headline)."""
    spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ks = [{'name': 'conv_gemm_x3h<256,128>', 'launches': 90, 'ms': 19.4, 'flops': 90 * 49.6e9},
          {'name': 'wgrad_x3r<3x3,2x64,128>', 'launches': 20, 'ms': 5.2, 'flops': 20 * 53e9},
          {'name': 'conv_pk<c3,k3>', 'launches': 48, 'ms': 1.0, 'flops': 48 * 0.23e9}]
    r = bench.roofline_of(ks, profile_steps=2, batch=64)
    assert r['kernel'] == 'conv_gemm_x3h<256,128>' and r['launches_per_step'] == 45 and r['peak'] == 2500.0
    alg = 90 * 49.6e9 / 19.4e-3 / 1e12
    assert abs(r['achieved'] - 6 * alg) < 0.01 and abs(r['frac'] - 6 * alg / 2500.0) < 1e-4
    assert abs(r['all_gemm_kernels']['ms_per_step'] - 12.8) < 1e-9 and r['traffic'] is None
    r2 = bench.roofline_of([{'name': 'conv_gemm<128,128,vec>', 'launches': 10, 'ms': 1.0, 'flops': 1e12}], 1, 64)
    assert r2['peak'] == 157.3 and abs(r2['achieved'] - 1000.0) < 1e-6            # exact-fp32 MFMA kernels: algorithmic flops, fp32 peak
    assert bench.roofline_of([], 1, 64) is None
    # side_leg: a child that prints one contract line ...
    child = tmp_path / 'child.py'
    child.write_text('import json,sys\nprint("noise")\nprint(json.dumps({"value": 3400.5, "unit": "images/s", "ms_per_step": 18.8, "steps": 10, "warmup": 3,'
                     ' "dtype": "bf16", "losses_finite": True, "config": {"per_gpu_batch": 64, "workload": "child text"},'
                     ' "roofline": {"kernel": "conv_bf16p<128,128>", "frac": 0.2, "traffic": 6.4e7}, "metric": "not kept"}))\n')
    monkeypatch.setattr(bench, '__file__', str(child))
    leg = bench.side_leg(['--dtype', 'bf16'], 'workload text', deadline=1e18, live=True)
    assert leg['value'] == 3400.5 and leg['roofline']['traffic'] == 6.4e7 and leg['config'] == {'workload': 'workload text', 'per_gpu_batch': 64}
    assert 'metric' not in leg and leg['command'] == 'python bench.py --dtype bf16'
    # ... and one that dies
    child.write_text('import sys\nsys.exit(3)\n')
    bad = bench.side_leg(['--dtype', 'bf16'], 'w', deadline=1e18, live=True)
    assert bad['value'] is None and 'error' in bad
