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
