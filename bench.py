#!/usr/bin/env python
"""Throughput bench of the EdgeGAN training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one training iteration = the 7 optimizer sub-steps of the reference's
EdgeGAN.update_model (edgegan/models/edgegan.py:109-130) on one synthetic batch that is already
resident in HBM.  Workload = BASELINE.json configs[2]: 14-class 64x128 fp32 training, batch 64 per
GPU; N > 1 is pure data parallel (weak scaling: 64 images per GPU, one RCCL all-reduce per optimised
network group).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 93.551 GFLOP of algorithmic conv/deconv/linear work per image and iteration (64x128,
# 14-class), minus the 2 generator forwards the engine legally shares across sub-steps 1-3 (2 x 0.3263)
GFLOP_PER_IMAGE = 93.551 - 2 * 0.3263
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (never the 2:1-sparse figure)
PMC_SET = 'r06y'                # the profile set profiles/pmc_utilisation_latest.csv was derived from (tools/gpu_round_profile.sh)
X3_LIMB_PRODUCTS = 6            # limb products issued per fp32 product in f32x3 mode (edgegan_amd/csrc/k_gemm_x3.hip)


def pmc_traffic(kernel_label):
    """HBM bytes per launch of `kernel_label` (e.g. 'conv_gemm<128,128,vec>', 'conv_gemm<128,128,vec,splitk>') from
    the committed rocprofv3 PMC summary (profiles/pmc_hbm_traffic_latest.csv: separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 x2 read correction -- tools/pmc_summary.py).  None if there is no matching row."""
    import csv
    import re
    # halo-staged kernels (round 2): per-kernel table derived by tools/pmc_derive.py from the round's PMC passes
    mh = re.match(r'conv_gemm_(x3h|bf16h)<256,(\d+)(,splitk)?>', kernel_label)
    upath = os.path.join(ROOT, 'profiles', 'pmc_utilisation_latest.csv')
    if mh and os.path.exists(upath):
        want = (6 if mh.group(1) == 'x3h' else 1, 'true' if mh.group(3) else 'false', mh.group(2))
        best = None
        with open(upath) as f:
            for r in csv.DictReader(f):
                mk = re.match(r'conv_gemm_x3h_kernel<(\d+), (true|false)(?:, (\d+))?', r['kernel'])
                if not mk or (int(mk.group(1)), mk.group(2)) != want[:2] or (mk.group(3) and mk.group(3) != want[2]):
                    continue
                if re.search(r', true>', r['kernel']):   # (sixth template argument: the phase-pair form, another bucket)
                    continue
                if best is None or int(r['launches']) > int(best['launches']):
                    best = r
        if best:
            return float(best['hbm_read_bytes_per_launch(2xFETCH_SIZE)']) + float(best['hbm_write_bytes_per_launch'])
    # plane-fed bf16 kernels (round 4): conv_bf16p<PX,BN[,splitk]> = conv_bf16p_kernel<MI, WN, SPLITK, ...> with
    # (PX, BN) -> (MI, WN): (128,128) -> (2,2), (256,128) -> (4,2), (256,64) -> (2,1); the bf16 round profile's table
    mp = re.match(r'conv_bf16p<(\d+),(\d+)(,splitk)?>', kernel_label)
    bpath = os.path.join(ROOT, 'profiles', 'pmc_utilisation_bf16_latest.csv')
    if mp and os.path.exists(bpath):
        mi_wn = {('128', '128'): ('2', '2'), ('256', '128'): ('4', '2'), ('256', '64'): ('2', '1')}.get((mp.group(1), mp.group(2)))
        best = None
        with open(bpath) as f:
            for r in csv.DictReader(f):
                mk = re.match(r'conv_bf16p_kernel<(\d+), (\d+), (true|false)', r['kernel'])
                if not mk or not mi_wn or (mk.group(1), mk.group(2)) != mi_wn or (mk.group(3) == 'true') != bool(mp.group(3)):
                    continue
                if best is None or int(r['launches']) > int(best['launches']):
                    best = r
        if best:
            return float(best['hbm_read_bytes_per_launch(2xFETCH_SIZE)']) + float(best['hbm_write_bytes_per_launch'])
    path = os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic_latest.csv')
    m = re.match(r'(\w+)<(\d+),(\d+)(?:,(\w+))?(?:,(splitk))?>', kernel_label)
    if not m or not os.path.exists(path):
        return None
    base, bm, bn, mode, splitk = m.groups()
    best = None
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r['Kernel']
            if f'{base}_kernel<{bm}, {bn},' not in k:
                continue
            flags = re.findall(r'(true|false)', k)
            want = {'vec': ['true', 'true'], 'avec': ['true', 'false'], 'flat': ['false', 'false']}.get(mode)
            if base == 'conv_gemm' and want and (flags[:2] != want or (flags[2:3] == ['true']) != bool(splitk)):
                continue
            if base == 'wgrad' and mode and flags[-1:] != (['true'] if mode == 'vec' else ['false']):
                continue
            if best is None or int(r['Launches']) > int(best['Launches']):
                best = r
    return float(best['HbmBytesPerLaunch']) if best else None


def live_traffic(kernel_label, args, mode='train'):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE:
    counters only, one group per pass, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over a 2-iteration child run of
    this same bench with the sub-steps serialised, averaged over that kernel's launches; read side x 2 (the guide's gfx950
    correction: FETCH_SIZE reports half of a wide coalesced read), both counters in KiB.  None where rocprofv3 is missing, the
    process already runs under a profiler, or the label has no known instantiation pattern -- the caller then falls back to
    the committed profile set."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    if not shutil.which('rocprofv3') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None
    m = re.match(r'conv_gemm_(x3h|bf16h)<256,(\d+)(,s2d)?(,splitk)?>', kernel_label)
    mp = re.match(r'conv_bf16p<(\d+),(\d+)(,splitk)?>', kernel_label)
    if m:
        # template arguments <NPROD, SPLITK, BN, SW16, S2D, PAIR>: the 2 x 64 phase-pair form (PAIR) has its own label and bucket
        pat = r'conv_gemm_x3h_kernel<%d, %s, %s, (true|false), %s, false>' % (6 if m.group(1) == 'x3h' else 1, 'true' if m.group(4) else 'false',
                                                                               m.group(2), 'true' if m.group(3) else 'false')
    elif mp:
        mi_wn = {('128', '128'): (2, 2), ('256', '128'): (4, 2), ('256', '64'): (2, 1)}.get((mp.group(1), mp.group(2)))
        if not mi_wn:
            return None
        pat = r'conv_bf16p_kernel<%d, %d, %s,' % (mi_wn[0], mi_wn[1], 'true' if mp.group(3) else 'false')
    else:
        return None
    rx = re.compile(pat)
    child = [sys.executable, os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--profile-steps', '0', '--no-cpu-baseline',
             '--no-alt', '--no-live-traffic', '--batch', str(args.batch), '--classes', str(args.classes), '--height', str(args.height),
             '--width', str(args.width), '--dtype', args.dtype, '--no-side-legs', '--mode', mode]
    env = dict(os.environ, EG_LANES='0', TMPDIR='/tmp')
    avg = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='eg_pmc_', dir='/tmp')
        try:
            subprocess.run(['rocprofv3', '--pmc', counter, '-d', d, '-o', 'p', '--output-format', 'csv', '--'] + child, env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90, check=True)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            n, tot = 0, 0.0
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get('Counter_Name') == counter and rx.search(r.get('Kernel_Name', '')):
                            n += 1
                            tot += float(r['Counter_Value'])
            if n == 0:
                return None
            avg[counter] = (tot / n, n)
        except (subprocess.SubprocessError, OSError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {'bytes': 2.0 * 1024.0 * avg['FETCH_SIZE'][0] + 1024.0 * avg['WRITE_SIZE'][0],
            'read_bytes': 2.0 * 1024.0 * avg['FETCH_SIZE'][0], 'write_bytes': 1024.0 * avg['WRITE_SIZE'][0],
            'launches_sampled': avg['FETCH_SIZE'][1]}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def physical_cores(hw):
    try:
        sib = set()
        for c in range(hw):
            with open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list') as f:
                sib.add(f.read().strip())
        return max(1, len(sib))
    except OSError:
        return hw


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU bandwidth limit (v2 cpu.max / v1 cfs quota) if there is one, else None.
    The GPU boxes of this pool run the job in a container limited to 16 CPUs on a 2 x 64-core host (`1600000 100000` in
    /sys/fs/cgroup/cpu.max, measured round 5): with more runnable threads than that the scheduler throttles the whole group,
    which is why "one thread per physical core" measured 0.7 images/s against 5.4 on 16 threads and why pinned shard processes
    slowed each other down 10-fold (profiles/r05e_cpu_scale.txt)."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _oracle_iteration(cfg_kwargs, batch, threads, substeps=None):
    """Seconds of one training iteration (or of the listed sub-steps) of the torch-CPU fp32 oracle at `batch`."""
    import torch
    import edgegan_amd as ea
    import edgegan_oracle as O
    import parity
    torch.set_num_threads(threads)
    kw = dict(cfg_kwargs)
    kw['batch_size'] = batch
    cfg = ea.Config(**kw)
    values = ea.init_variables(cfg, 0)
    b = ea.synthetic_batch(cfg)
    om = O.OracleModel(parity.oracle_config(cfg), values, dtype=torch.float32)
    inp = O.StepInputs(*[torch.tensor(b[k]) for k in ('images', 'z', 'alpha', 'eps')])
    t0 = time.time()
    for name in (substeps or om.substep_names()):
        om.run_substep(name, inp)
    return time.time() - t0


def _cpu_shard_worker(rank, cores, threads, cfg_kwargs, batch, barrier, out):
    """One process of the all-core CPU baseline: pinned to its own cores, its own shard of the batch."""
    try:
        if cores:
            os.sched_setaffinity(0, cores)
        os.environ['OMP_NUM_THREADS'] = str(threads)
        for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch
        torch.set_num_interop_threads(1)
        _oracle_iteration(cfg_kwargs, batch, threads)      # untimed: thread pool, allocator, oneDNN primitives of these shapes
        barrier.wait()
        out.put((rank, _oracle_iteration(cfg_kwargs, batch, threads)))
    except Exception as e:     # the parent must never wait for a worker that died
        try:
            barrier.abort()
        except Exception:
            pass
        out.put((rank, repr(e)))


def cpu_all_cores_sharded(cfg_kwargs, batch_all, threads, phys):
    """All physical cores the way a CPU data-parallel run would use them: P = cores // threads processes, each pinned to its
    own `threads` cores with 1/P of the batch (one torch process does not scale past ~16 threads on these convolution
    shapes; one thread per core in ONE process thrashes its spin-waiting pool).  Returns (images/s, P, per-process batch, s)."""
    import multiprocessing as mp
    procs = max(1, phys // threads)
    per = max(2, batch_all // procs)
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(phys))
    ctx = mp.get_context('spawn')
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = []
    for r in range(procs):
        cores = set(avail[r * threads:(r + 1) * threads]) if len(avail) >= procs * threads else None
        ps.append(ctx.Process(target=_cpu_shard_worker, args=(r, cores, threads, cfg_kwargs, per, barrier, out)))
        ps[-1].start()
    import queue
    times, waited = [], 0.0
    while len(times) < procs:
        try:
            r, t = out.get(timeout=2.0)
        except queue.Empty:
            waited += 2.0
            if waited > 600 or any(p.exitcode not in (None, 0) for p in ps):     # a worker died without reporting / hung
                for p in ps:
                    p.terminate()
                raise RuntimeError('cpu shard workers did not report (exit codes %s)' % [p.exitcode for p in ps])
            continue
        if not isinstance(t, float):
            for p in ps:
                p.terminate()
            raise RuntimeError(f'cpu shard {r}: {t}')
        times.append(t)
    for p in ps:
        p.join(timeout=60)
    return procs * per / max(times), procs, per, max(times)


def cpu_baseline(cfg_kwargs, batch_all=64, batch_one=2):
    """the CPU restatement of the identical iteration (`kind: port`, torch-CPU -- the reference's
    TF-1.14 CPU path cannot exist on this box), timed on this host's cores:
      (a) `value`: one full iteration (7 sub-steps) at the bench batch (configs[2] shapes) with the BEST thread count of a
          short sweep at batch 16 -- torch-CPU does not scale to all hardware threads on these conv shapes, so "all
          cores" alone would understate the CPU; `cores` is the thread count actually used;
      (b) `all_cores`: the same iteration (batch 8) with one thread on EVERY physical core of the box (printed although it
          is worse than (a));
      (c) `single_core`: ONE thread, a full iteration at batch `batch_one`;
      (d) `config0_single_class_b4`: BASELINE configs[0]'s shape (single-class, batch 4: the reference's own CPU-runnable
          case) at the best thread count.
    torch-CPU fp32, NOT the reference's TensorFlow."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    ncpu = torch.get_num_threads()
    hw = os.cpu_count() or ncpu
    quota = cpu_quota()
    # every CPU the process may use: the host's physical cores, or the container's CPU quota where that is smaller (then a
    # thread count above it only gets the whole group throttled -- cpu_quota())
    phys = physical_cores(hw)
    usable = phys if quota is None else max(1, min(phys, int(quota)))
    _oracle_iteration(cfg_kwargs, 2, min(ncpu, 16, usable), ['e_optim'])          # untimed: thread pool / allocator warm-up
    sweep = {th: round(16 / _oracle_iteration(cfg_kwargs, 16, th), 2) for th in sorted({8, 16, 32, 64, usable}) if th <= usable} or \
        {usable: round(16 / _oracle_iteration(cfg_kwargs, 16, usable), 2)}
    best = max(sweep, key=sweep.get)
    # "all cores" = every USABLE core, one thread each (with SMT siblings added torch's spin-waiting pool thrashes)
    dt_allc = _oracle_iteration(cfg_kwargs, 8, usable) if usable not in sweep else 16 / max(sweep[usable], 1e-9)
    allc_batch = 8 if usable not in sweep else 16
    dt_all = _oracle_iteration(cfg_kwargs, batch_all, best)
    dt_one = _oracle_iteration(cfg_kwargs, batch_one, 1)
    dt_c0 = _oracle_iteration(dict(cfg_kwargs, num_classes=0), 4, best)
    torch.set_num_threads(ncpu)
    # the best HONEST all-core number: batch shards on pinned processes
    sharded = None
    if usable >= 2 * best:
        try:
            ips, procs, per, dts = cpu_all_cores_sharded(cfg_kwargs, batch_all, best, usable)
            sharded = {'value': round(ips, 3), 'unit': 'images/s', 'cores': procs * best,
                       'sample': f'{procs} processes x {best} threads, each pinned to its own cores, one full iteration at batch '
                                 f'{per} per process ({procs * per} images), {dts:.1f} s for the slowest'}
        except Exception as e:
            sharded = {'value': None, 'error': repr(e)}
    if sharded and sharded.get('value') and sharded['value'] > batch_all / dt_all:
        head = {'value': sharded['value'], 'cores': sharded['cores'], 'sample': sharded['sample'] + ' (torch-CPU fp32 oracle, BASELINE configs[2] shapes)'}
    else:
        head = {'value': round(batch_all / dt_all, 3), 'cores': best,
                'sample': f'1 full iteration (7 sub-steps) at batch {batch_all} (BASELINE configs[2] shapes), torch-CPU fp32 '
                          f'oracle, {best} threads (best of the sweep), {dt_all:.1f} s'}
    return {'value': head['value'], 'unit': 'images/s', 'cores': head['cores'], 'kind': 'port',
            'kind_detail': 'torch-CPU fp32 restatement (oracle/edgegan_oracle.py), not the reference\'s TensorFlow 1.14',
            'cpu_model': cpu_model(), 'host_hardware_threads': hw, 'host_physical_cores': phys,
            'cgroup_cpu_quota': quota, 'usable_cpus': usable,
            'usable_cpus_note': ('the container is limited to %g CPUs by its cgroup CPU quota: `cores` below are ALL the CPUs this '
                                 'process may use, not a subset of the host\'s %d' % (quota, phys)) if quota is not None and quota < phys
            else 'no cgroup CPU quota below the host\'s physical core count',

            'sample': head['sample'],
            'one_process_best_threads': {'value': round(batch_all / dt_all, 3), 'unit': 'images/s', 'cores': best,
                                         'sample': f'1 full iteration at batch {batch_all}, {best} threads, {dt_all:.1f} s'},
            'all_cores_sharded': sharded,
            'thread_sweep_images_per_s_at_batch_16': sweep,
            'all_cores': {'value': round(allc_batch / dt_allc, 3), 'unit': 'images/s', 'cores': usable,
                          'sample': f'1 full iteration at batch {allc_batch} with one thread on each of the {usable} usable CPUs '
                                    f'(host: {phys} physical cores, {hw} hardware threads; cgroup quota {quota}), {dt_allc:.1f} s'},
            'single_core': {'value': round(batch_one / dt_one, 4), 'unit': 'images/s', 'cores': 1,
                            'sample': f'1 full iteration at batch {batch_one}, 1 thread, {dt_one:.1f} s'},
            'config0_single_class_b4': {'value': round(4 / dt_c0, 3), 'unit': 'images/s', 'cores': best,
                                        'sample': f'BASELINE configs[0]: single-class 64x128, batch 4, 1 full iteration '
                                                  f'(6 sub-steps: no classifier), {dt_c0:.1f} s'}}


def alt_mode(ea, cfg_kwargs, values, batch, args):
    """The same iteration with the dense contractions on the exact-fp32 MFMA instruction (compute_dtype 'f32'), timed in
    the same process right after the main measurement: the two fp32 implementations side by side."""
    cfg = ea.Config(**cfg_kwargs)
    n2 = ea.EdgeGAN(cfg, values)
    n2.upload_batch(batch['images'], batch['z'], batch['alpha'], batch['eps'])
    steps = max(5, args.steps // 2)
    for _ in range(2):
        n2.train_step()
    n2.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        n2.train_step()
    n2.sync()
    dt = time.perf_counter() - t0
    n2.close()
    return {'value': round(args.batch * steps / dt, 2), 'unit': 'images/s', 'ms_per_step': round(dt / steps * 1e3, 3),
            'steps': steps, 'note': 'dense contractions on v_mfma_f32_16x16x4_f32 (python bench.py --dtype f32)'}


INFER_GFLOP_PER_IMAGE = 2.228     # E (1.575) + G1 + G2 (2 x 0.326), 64x128


def roofline_of(ks, profile_steps, batch, note=None):
    """`roofline` object from the engine's per-kernel event records (eg_profile_end: every GEMM-shaped launch bracketed by HIP
    events on the stream it runs on, sub-steps serialised): the dominant kernel = the bucket with the most time."""
    if not ks:
        return None
    dom = max(ks, key=lambda k: k['ms'])
    tot_ms = sum(k['ms'] for k in ks)
    tot_fl = sum(k['flops'] for k in ks)
    alg = dom['flops'] / (dom['ms'] * 1e-3) / 1e12       # algorithmic fp32 (or bf16) TFLOP/s of the kernel
    limb = 'x3' in dom['name']         # limb kernels execute 6 bf16 MFMA flops per algorithmic fp32 flop
    ach = alg * (X3_LIMB_PRODUCTS if limb else 1)
    bf = limb or 'bf16' in dom['name']
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
    r = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
         'traffic': None, 'kernel': dom['name'], 'launches_per_step': dom['launches'] / profile_steps,
         'avg_launch_ms': round(dom['ms'] / dom['launches'], 4), 'flops_per_launch': dom['flops'] / dom['launches'],
         'all_gemm_kernels': {'tflops': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), 'ms_per_step': round(tot_ms / profile_steps, 3),
                              'frac_of_peak': round(tot_fl / (tot_ms * 1e-3) / 1e12 * (X3_LIMB_PRODUCTS if limb else 1) / peak, 4),
                              'executed_gflop_per_image': round(tot_fl / profile_steps / batch / 1e9, 2)}}
    if note:
        r['note'] = note
    return r


def attach_live_traffic(roof, spec, mode, deadline):
    """roofline.traffic of a side leg, measured like the headline's (two rocprofv3 --pmc child runs); skipped -- and said so --
    once the whole bench has used its time budget."""
    if not roof:
        return
    if time.time() > deadline:
        roof['traffic_source'] = 'not measured: the run had used its time budget before this leg\'s PMC passes'
        return
    lt = live_traffic(roof['kernel'], spec, mode)
    if lt:
        roof['traffic'] = round(lt['bytes'], 1)
        roof['traffic_read_bytes'] = round(lt['read_bytes'], 1)
        roof['traffic_write_bytes'] = round(lt['write_bytes'], 1)
        roof['traffic_source'] = ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over a child run of this '
                                  'leg with the launches serialised, averaged over %d launches of this kernel; read side x 2 (gfx950), '
                                  'KiB units' % lt['launches_sampled'])


def side_leg(extra_args, workload, deadline, live, timeout=150):
    """One more BASELINE configuration in the default run's line: this same bench in a FRESH process (own HIP context, no torch
    threads of the CPU baseline around, the same state a stand-alone `python bench.py <extra_args>` has -- in-process legs
    measured 4 ... 25 % below their stand-alone runs), its one JSON line reduced to what the leg reports."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_args + ['--no-cpu-baseline', '--no-alt', '--no-side-legs']
    if not live or time.time() > deadline:
        cmd.append('--no-live-traffic')
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ))
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        d = json.loads(line)
    except Exception as e:      # a side leg must never hide the headline
        return {'value': None, 'error': repr(e)}
    keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'losses_finite', 'outputs_finite', 'memory', 'host_enqueue_ms_per_step',
            'algorithmic_tflops_per_gpu', 'roofline')
    leg = {k: d[k] for k in keep if k in d}
    leg['config'] = {'workload': workload, 'per_gpu_batch': d.get('config', {}).get('per_gpu_batch')}
    leg['command'] = 'python bench.py ' + ' '.join(extra_args)
    if '--no-live-traffic' in cmd and live and leg.get('roofline'):
        leg['roofline']['traffic_source'] = 'not measured: the run had used its time budget before this leg\'s PMC passes'
    return leg


def infer_leg(ea, np, rank, world, local, spec, steps, warmup, profile_steps, deadline, live):
    """BASELINE configs[1]: single-class 64x128, batch 64, generator-only inference (edgegan.test: E -> G1, E -> G2 = the
    two sess.run calls of edgegan/models/edgegan.py:606-609), bf16 contractions, inputs resident in HBM.  A step = one
    batch through both runs.  N > 1: independent replicas (inference has no exchange step)."""
    cfg = ea.Config(batch_size=spec.batch, num_classes=spec.classes, device=local, compute_dtype=spec.dtype, height=spec.height,
                    width=spec.width)
    net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
    if world > 1:
        from edgegan_amd import rendezvous
        rendezvous.init_data_parallel(net, rank, world)
    b = ea.synthetic_batch(cfg, seed=1234, rank=rank)
    cls = (np.arange(spec.batch) % max(cfg.num_classes, 1)).astype(np.int32)
    net.infer_upload(b['images'], cls, eps2=[0.1, 0.2])

    def sync():
        net.sync()
        net.comm_barrier()
        net.sync()
    for _ in range(warmup):
        net.infer_run()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.infer_run()
    t_enq = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = float(net.comm_reduce_scalars([dt], 'max')[0])
    roof = None
    if rank == 0 and profile_steps > 0:
        net.profile_begin()
        for _ in range(profile_steps):
            net.infer_run()
        roof = roofline_of(net.profile_end()['kernels'], profile_steps, spec.batch,
                           note='a batch-64 forward of E, G1, G2: small maps (4x4 .. 64x64), bounded by launch count and per-launch '
                                'latency before it is bounded by the matrix pipe')
    e, i = net.infer_download()
    finite = bool(np.all(np.isfinite(e)) and np.all(np.isfinite(i)))
    net.close()
    if live and rank == 0 and world == 1:
        attach_live_traffic(roof, spec, 'infer', deadline)
    value = spec.batch * world * steps / dt
    return {'value': round(value, 1), 'unit': 'images/s', 'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'warmup': warmup,
            'dtype': spec.dtype,
            'config': {'workload': f'configs[1]: {"single-class" if cfg.num_classes == 0 else str(cfg.num_classes) + "-class"} '
                                   f'{spec.height}x{spec.width} {spec.dtype} generator-only inference (edgegan.test path), batch '
                                   f'{spec.batch} per GPU, inputs resident in HBM', 'per_gpu_batch': spec.batch,
                       'parallelism': f'replicas x{world}'},
            'algorithmic_tflops_per_gpu': round(INFER_GFLOP_PER_IMAGE * value / world / 1e3, 2) if (spec.height, spec.width) == (64, 128) else None,
            'host_enqueue_ms_per_step': round(t_enq / steps * 1e3, 3), 'outputs_finite': finite, 'roofline': roof}



def infer_bench(args, ea, np, rank, world, local):
    """python bench.py --mode infer: the configs[1] leg on its own (one JSON line in the contract's shape)."""
    from types import SimpleNamespace
    spec = SimpleNamespace(batch=args.batch, classes=0 if '--classes' not in sys.argv else args.classes, height=args.height,
                           width=args.width, dtype='bf16' if '--dtype' not in sys.argv else args.dtype)
    leg = infer_leg(ea, np, rank, world, local, spec, args.steps, args.warmup, args.profile_steps, time.time() + 120,
                    live=not args.no_live_traffic)
    if rank == 0:
        out = {'metric': f'inference images/sec (E -> G1, G2) at {args.height}x{args.width}', 'value': leg['value'], 'unit': 'images/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': leg['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': spec.dtype, 'data': 'synthetic'}
        out.update({k: v for k, v in leg.items() if k not in out})
        print(json.dumps(out))


def loop_bench(args, ea, np, local):
    """End-to-end throughput of the REAL training loop (edgegan/models/edgegan.py:456-489 as python -m edgegan_amd.train
    runs it) beside the resident-batch number: PNG files on disk -> Dataset / Prefetcher (edgegan/utils/data/dataset.py:
    58-69 semantics: decode, bytescale, resize, /127.5-1) -> host->device feed -> the 7 sub-steps -> per-iteration loss
    read-back.  Measured in one process, same engine: (a) resident batch (what `value` of the default mode measures),
    (b) the pipelined loop (update_model_async + losses one iteration late: what train.py does), (c) the synchronous
    loop of round 2 (update_model, then losses(): a host<->device synchronisation per iteration), (d) the loader alone."""
    import shutil
    import tempfile
    from PIL import Image
    from edgegan_amd.data import Dataset, Prefetcher
    K, W = args.steps, args.warmup
    cfg = ea.Config(batch_size=args.batch, num_classes=args.classes, device=local, compute_dtype=args.dtype,
                    height=args.height, width=args.width)
    root = tempfile.mkdtemp(prefix='edgegan_loop_')
    try:
        rng = np.random.RandomState(0)
        n_img = (K + W) * args.batch
        ncls = max(args.classes, 1)
        t0 = time.perf_counter()
        for c in range(ncls):
            d = os.path.join(root, 'synth', 'train', str(c)) if args.classes else os.path.join(root, 'synth', 'train')
            os.makedirs(d, exist_ok=True)
            for i in range((n_img + ncls - 1) // ncls):
                # smooth random fields (photo-like compressibility), not white noise
                low = rng.randint(0, 256, size=(args.height // 8, args.width // 8, 3)).astype(np.uint8)
                Image.fromarray(low).resize((args.width, args.height), Image.BILINEAR).save(os.path.join(d, f'{c}_{i}.png'))
        t_make = time.perf_counter() - t0
        dcfg = {'input_height': args.height, 'input_width': args.width, 'output_height': args.height,
                'output_width': args.width, 'crop': False, 'grayscale': False, 'z_dim': cfg.z_dim}
        ds = Dataset(root, 'synth', float('inf'), args.batch, dcfg, args.classes or None, 'train')
        ds.shuffle()
        net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
        order = list(range(K + W))
        # (d) loader alone
        t0 = time.perf_counter()
        n = 0
        for _ in Prefetcher(ds, order):
            n += 1
        loader_ips = n * args.batch / (time.perf_counter() - t0)
        # (a) resident batch
        b = ea.synthetic_batch(cfg, seed=1234)
        net.upload_batch(b['images'], b['z'], b['alpha'], b['eps'])
        for _ in range(W):
            net.train_step()
        net.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            net.train_step()
        net.sync()
        resident = args.batch * K / (time.perf_counter() - t0)

        def run(pipelined):
            t0 = None
            late = False
            for it, (idx, images, z, _) in enumerate(Prefetcher(ds, order)):
                if it == W:
                    net.sync()
                    t0 = time.perf_counter()
                if pipelined:
                    net.update_model_async(images, z)
                    if late:
                        l = net.losses_async(back=1)
                    late = True
                else:
                    net.update_model(images, z)
                    l = net.losses()
                if it >= 1 or not pipelined:
                    _ = '%.8f %.8f' % (2 * (l['joint_dis_dloss'] + l['image_dis_dloss'] + l['edge_dis_dloss']), l['edge_gloss'] + l['image_gloss'])
            if pipelined:
                l = net.losses_async(back=0)
            net.sync()
            return args.batch * K / (time.perf_counter() - t0), l
        piped, l1 = run(True)
        synced, l2 = run(False)
        print(json.dumps({
            'metric': f'training images/sec (G+D step) at {args.height}x{args.width}, 14-class -- END-TO-END LOOP', 'unit': 'images/s',
            'value': round(piped, 2), 'n_gpus': 1, 'steps': K, 'warmup': W, 'dtype': args.dtype, 'data': 'synthetic PNG files on local disk',
            'higher_is_better': True,
            'config': {'workload': f'python -m edgegan_amd.train loop: PNG decode -> feed -> 7 sub-steps -> loss read-back, batch {args.batch}'},
            'resident_batch_images_per_s': round(resident, 2), 'loop_pipelined_images_per_s': round(piped, 2),
            'loop_synchronous_images_per_s': round(synced, 2), 'loader_alone_images_per_s': round(loader_ips, 1),
            'loop_vs_resident': round(piped / resident, 4), 'decode_workers': Prefetcher(ds, []).workers,
            'host_hardware_threads': os.cpu_count(), 'png_tree_seconds': round(t_make, 1),
            'losses_finite': bool(all(np.isfinite(v) for v in l1.values()) and all(np.isfinite(v) for v in l2.values()))}))
        net.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command line (one process per GPU, as
    the reference is one process per device, edgegan/train.py:77), with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a
    launch-unique EG_RDZV_KEY in their environment (the ranks' RCCL id travels through edgegan_amd/rendezvous.py).  Rank
    0's stdout -- the ONE JSON line -- is passed through; a rank that fails takes the others down with it."""
    import secrets
    import socket
    import subprocess
    with socket.socket() as sk:          # a free port: part of the rendezvous key and what torch-style tools expect
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    key = os.environ.get('EG_RDZV_KEY') or f'bench{os.getpid()}_{secrets.token_hex(4)}'
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), EG_RDZV_KEY=key)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f'bench.py: rank {r} exited with {code}; stopping the other ranks', file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser(epilog='N > 1: either plain `python bench.py --gpus N` (this process then starts the N ranks itself) or '
                                        '`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` '
                                        '(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  The RCCL id travels through a '
                                        'file keyed on the launcher process; a launcher whose ranks do not share a parent process must '
                                        'export the same EG_RDZV_KEY to every rank.  EG_RCCL_LIB=<path> binds a specific librccl.')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--classes', type=int, default=14)
    ap.add_argument('--height', type=int, default=64, help='64 (configs[2], the metric) or 256 (configs[4] geometry)')
    ap.add_argument('--width', type=int, default=128)
    ap.add_argument('--profile-steps', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the side measurement of the exact-fp32-MFMA mode')
    ap.add_argument('--no-side-legs', action='store_true',
                    help='skip the side legs of the default run (bf16_mode = configs[3] arithmetic at batch 64, '
                         'config4_256x512_b16_bf16, config1_infer_b64), each with its own roofline object')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='roofline.traffic from the committed profile set instead of two rocprofv3 --pmc passes inside this run')
    ap.add_argument('--cpu-baseline-batch', type=int, default=64)
    ap.add_argument('--mode', choices=['train', 'infer', 'loop'], default='train',
                    help='train = BASELINE configs[2] (the headline metric); infer = configs[1]: single-class 64x128 bf16 '
                         'generator-only inference (edgegan.test path E -> G1, E -> G2), batch 64, inputs resident in HBM; '
                         'loop = the real training loop end to end (PNG files -> loader -> feed -> iteration -> loss read-back) '
                         'beside the resident-batch number, one GPU')
    ap.add_argument('--dtype', choices=['f32x3', 'f32', 'bf16'], default='f32x3',
                    help="f32x3 (default) and f32 = BASELINE configs[2] (the headline metric) in fp32: f32x3 forms every fp32 "
                         "product of the dense contractions from exact bf16-limb products on the bf16 MFMA (fp32 accumulate), "
                         "f32 uses the exact-fp32 MFMA instruction; bf16 = configs[3]'s arithmetic: bf16 MFMA "
                         "operands with fp32 accumulation / storage")
    args = ap.parse_args()
    t_start = time.time()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if 'EG_FORCE_DEVICE' in os.environ:      # test hook: several ranks on one GPU (1-GPU dev boxes)
        local = int(os.environ['EG_FORCE_DEVICE'])
    if world != args.gpus:
        if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N` (no launcher): this process becomes the launcher of N ranks
            sys.exit(self_launch(args.gpus))
        args.gpus = world

    import numpy as np
    import edgegan_amd as ea

    if args.mode == 'infer':
        if '--steps' not in sys.argv:       # (a batch takes < 0.5 ms: enough of them for the calls in flight to reach their steady state)
            args.steps, args.warmup = 400, 20
        return infer_bench(args, ea, np, rank, world, local)
    if args.mode == 'loop':
        if '--steps' not in sys.argv:
            args.steps = 50
        return loop_bench(args, ea, np, local)

    cfg_kwargs = dict(batch_size=args.batch, num_classes=args.classes, device=local, compute_dtype=args.dtype,
                      height=args.height, width=args.width)
    # algorithmic GFLOP per image and iteration; both minus the two shared generator forwards
    gflop_per_image = {(64, 128): GFLOP_PER_IMAGE, (256, 512): 1102.3 - 2 * 5.22}.get((args.height, args.width))
    cfg = ea.Config(**cfg_kwargs)
    values = ea.init_variables(cfg, 0)                 # same weights on every rank
    net = ea.EdgeGAN(cfg, values)
    batch = ea.synthetic_batch(cfg, seed=1234, rank=rank)   # each rank its own shard of the global batch
    if world > 1:       # the 128-byte RCCL id travels through a file; every later exchange runs over RCCL itself
        from edgegan_amd import rendezvous
        rendezvous.init_data_parallel(net, rank, world)
    net.upload_batch(batch['images'], batch['z'], batch['alpha'], batch['eps'])

    def barrier_sync():
        # (eg_sync = hipDeviceSynchronize behind the engine's pending exchanges: what torch.cuda.synchronize() is to a torch
        #  program -- this process never loads torch)
        net.sync()
        net.comm_barrier()          # RCCL all-reduce of one scalar + device sync: returns when every rank has arrived
        net.sync()

    for _ in range(args.warmup):
        net.train_step()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.train_step()
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the K iterations (the stream runs behind)
    barrier_sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = float(net.comm_reduce_scalars([dt], 'max')[0])       # MAX over ranks
    losses = net.losses()
    finite = all(np.isfinite(v) for v in losses.values())

    # roofline of the dominant kernel: HIP events around every GEMM-shaped launch on the engine's stream.  While the
    # profiler is on, train_step runs its sub-steps one after another on one stream (a kernel is measured in isolation;
    # the timed region above overlaps independent sub-steps on several streams, which stretches individual kernels)
    # per-collective timings of the exchange (world > 1): two more iterations with the lanes ON -- the schedule of the timed
    # region -- and an event pair around every gradient all-reduce on the communication stream; per group the mean over the two
    # iterations, then the MAX over the ranks: a bad scaling line names the collective that cost it
    exchange_ms = None
    if world > 1:
        net.comm_timing(True)
        for _ in range(2):
            net.train_step()
        net.sync()
        recs = net.comm_timing_read()
        net.comm_timing(False)
        order = []
        acc = {}
        for i, (g, nbytes, ms) in enumerate(recs):
            key = f'{i % max(1, len(recs) // 2)}:{g}'          # (position in the iteration: the generators exchange twice)
            if key not in acc:
                order.append(key)
                acc[key] = [0.0, nbytes, 0]
            acc[key][0] += ms
            acc[key][2] += 1
        means = [acc[k][0] / acc[k][2] for k in order]
        if means:
            mx = net.comm_reduce_scalars(means, 'max')
            mn = net.comm_reduce_scalars(means, 'min')
            exchange_ms = [{'collective': k, 'mbytes': round(acc[k][1] / 1e6, 2), 'ms_max_over_ranks': round(float(mx[i]), 3),
                            'ms_min_over_ranks': round(float(mn[i]), 3),
                            'algbw_GBps_at_max': round(acc[k][1] / 1e9 / max(float(mx[i]) * 1e-3, 1e-9), 1)} for i, k in enumerate(order)]
    roof = None
    if args.profile_steps > 0:
        # EVERY rank runs the profiled iterations: under data parallelism a training iteration issues the groups'
        # all-reduces, and a collective that only rank 0 enqueued would pair with the other ranks' next one
        net.profile_begin()
        for _ in range(args.profile_steps):
            net.train_step()
        prof = net.profile_end()
        ks = prof['kernels'] if rank == 0 else None
        if ks:
            dom = max(ks, key=lambda k: k['ms'])
            tot_ms = sum(k['ms'] for k in ks)
            tot_fl = sum(k['flops'] for k in ks)
            alg = dom['flops'] / (dom['ms'] * 1e-3) / 1e12       # algorithmic fp32 (or bf16) TFLOP/s of the kernel
            limb = 'x3' in dom['name']         # limb kernels execute 6 bf16 MFMA flops per algorithmic fp32 flop
            ach = alg * (X3_LIMB_PRODUCTS if limb else 1)
            peak = PEAK_BF16_MFMA_TFLOPS if ('bf16' in dom['name'] or limb) else PEAK_FP32_MFMA_TFLOPS
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4),
                    'achieved_is': ('executed bf16-limb MFMA flops = 6 x the algorithmic fp32 flops of the launch, against the '
                                    'dense bf16 MFMA peak' if limb else 'algorithmic flops of the launch'),
                    'algorithmic_tflops': round(alg, 2),
                    'algorithmic_frac_of_fp32_mfma_peak': round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                    'traffic': pmc_traffic(dom['name']) if (args.height, args.width, args.batch) == (64, 128, 64) else None,
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC passes FETCH_SIZE x 2 + WRITE_SIZE)',
                    'traffic_source': 'committed profile set ' + PMC_SET + ' (profiles/pmc_utilisation_latest.csv), not re-measured in this run',
                    'kernel': dom['name'],
                    'measured': 'HIP events, sub-steps serialised (EG_LANES=0 semantics) so that kernels do not overlap',
                    'launches_per_step': dom['launches'] / args.profile_steps,
                    'avg_launch_ms': round(dom['ms'] / dom['launches'], 4),
                    'flops_per_launch': dom['flops'] / dom['launches'],
                    'all_gemm_kernels': {'tflops': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                         'frac_of_peak': round(tot_fl / (tot_ms * 1e-3) / 1e12 * (X3_LIMB_PRODUCTS if limb else 1) / peak, 4),
                                         'ms_per_step': round(tot_ms / args.profile_steps, 3),
                                         'executed_gflop_per_image': round(tot_fl / args.profile_steps / args.batch / 1e9, 2)},
                    'per_kernel': {k['name']: {'ms_per_step': round(k['ms'] / args.profile_steps, 3),
                                               'tflops': round(k['flops'] / (k['ms'] * 1e-3) / 1e12, 2)} for k in ks}}
            if peak == PEAK_BF16_MFMA_TFLOPS:
                # what the matrix pipe of THIS device sustains on a memory-free bf16 MFMA loop, measured now: `peak` stays
                # the guide's nominal figure; with random operands the power management clocks the pipe lower
                sus_c, sus_r = net.mfma_probe(20000, False), net.mfma_probe(20000, True)
                roof['sustained_mfma_loop'] = {'constant_operands_tflops': round(sus_c, 1), 'random_operands_tflops': round(sus_r, 1),
                                               'frac_of_random_operand_rate': round(ach / sus_r, 4),
                                               'what': 'eg_mfma_probe: one 8-wave block per CU, 4 independent 32x32x16 bf16 MFMA chains per wave, no memory traffic'}
    net.comm_barrier()
    ci = net.comm_info()
    iters_run = args.warmup + args.steps + (args.profile_steps if args.profile_steps > 0 else 0) + (2 if world > 1 else 0)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {
            'metric': f'training images/sec (G+D step) at {args.height}x{args.width}, 14-class',
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'arithmetic': {'f32x3': 'fp32 storage / accumulation / optimizer; each fp32 operand of a dense contraction is split '
                                    'EXACTLY into three bf16 limbs and the product is formed from 6 exact limb products on '
                                    'v_mfma_f32_32x32x16_bf16 (dropped terms < 2^-23 of the product) -- forward, input-gradient AND '
                                    'filter-gradient kernels; thin layers (<= 16 channels on either side), N=1 heads, norms, '
                                    'activations, losses, RMSProp: fp32 instructions',
                           'f32': 'fp32 everywhere; dense contractions on v_mfma_f32_16x16x4_f32 (exact fp32 products)',
                           'bf16': 'operands of dense contractions ROUNDED to bf16, fp32 accumulation / storage'}[args.dtype],
            'config': {'workload': ((({'f32': 'configs[2]: 14-class 64x128 fp32 (exact-fp32 MFMA)',
                                       'f32x3': 'configs[2]: 14-class 64x128 fp32 (f32x3: fp32 results from exact bf16-limb products)'}[args.dtype]
                                      if args.dtype in ('f32', 'f32x3') else
                                      'configs[3] arithmetic (bf16 MFMA operands, fp32 accumulate/storage): 14-class 64x128')
                                     if (args.height, args.width) == (64, 128) else
                                     f'configs[4] geometry: 14-class {args.height}x{args.width} {args.dtype}')) +
                                   ' full G+D+classifier+E training iteration '
                                   f'(7 RMSProp sub-steps), batch {args.batch} per GPU', 'global_batch': args.batch * world,
                       'per_gpu_batch': args.batch, 'num_classes': args.classes, 'parallelism': f'dp{world}'},
            'algorithmic_tflops_per_gpu': round(gflop_per_image * value / world / 1e3, 2) if gflop_per_image else None,
            'algorithmic_frac_of_fp32_mfma_peak': round(gflop_per_image * value / world / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)
            if gflop_per_image else None,
            'algorithmic_frac_note': 'algorithmic fp32 FLOP/s of the whole iteration over the exact-fp32 MFMA peak (157.3 TF); in '
                                     'f32x3 mode the work executes on the bf16 pipe, so this is a speed-up figure, not a utilisation',
            'exchange': {'nranks': ci['nranks'], 'rccl_version': ci['rccl_version'], 'rccl_library': ci['rccl_library'],
                         'allreduce_calls_per_step': round(ci['allreduce_calls'] / iters_run, 2) if world > 1 else 0,
                         'allreduce_mbytes_per_step': round(ci['allreduce_bytes'] / iters_run / 1e6, 2) if world > 1 else 0.0,
                         'per_collective': exchange_ms,
                         'per_collective_what': 'issue order within an iteration; ms between the communication stream reaching the '
                                                'all-reduce (this rank\'s gradients final) and its completion (includes the wait for the slowest '
                                                'rank), mean of two iterations with the lanes on, max / min over the ranks',
                         'env': {k: os.environ[k] for k in ('EG_COMM_CHANNELS', 'EG_COMM_CUS', 'NCCL_MAX_NCHANNELS',
                                                            'NCCL_MIN_NCHANNELS', 'EG_LANES') if k in os.environ},
                         'what': 'one RCCL all-reduce (sum, fp32, in place) of the flat gradient buffer per optimised network '
                                 'group and sub-step (G1+G2 in one call), on a side stream, averaged inside RMSProp'},
            'losses_finite': bool(finite), 'memory': net.memory_stats(),
            'host_enqueue_ms_per_step': round(t_enq / args.steps * 1e3, 3),
            'roofline': roof,
        }
        if world == 1 and args.dtype == 'f32x3' and not args.no_alt:
            out['exact_fp32_mfma_mode'] = alt_mode(ea, dict(cfg_kwargs, compute_dtype='f32'), values, batch, args)
        if world == 1 and net is not None:
            net.close()         # (the side legs and the PMC child runs own contexts of their own)
            net = None
        if (world == 1 and not args.no_side_legs and args.dtype == 'f32x3'
                and (args.height, args.width, args.batch, args.classes) == (64, 128, 64, 14)):
            # the other BASELINE configurations, in the driver's one line: each leg = this bench in a fresh
            # process -- its own timed region (barrier + device sync on both sides), its own roofline object with the dominant
            # kernel's HBM traffic measured in ITS run.  PMC passes of a leg are skipped once the whole run is past its budget.
            deadline = t_start + 100
            live = not args.no_live_traffic
            out['bf16_mode'] = side_leg(
                ['--dtype', 'bf16', '--steps', '10', '--warmup', '3'],
                'configs[3] arithmetic (bf16 MFMA operands, fp32 accumulate / storage): 14-class 64x128 full training iteration '
                '(7 RMSProp sub-steps), batch 64 per GPU (the per-GPU workload of the 8-GPU DP configuration)', deadline, live)
            out['config4_256x512_b16_bf16'] = side_leg(
                ['--dtype', 'bf16', '--height', '256', '--width', '512', '--batch', '16', '--steps', '5', '--warmup', '2', '--profile-steps', '1'],
                'configs[4]: 14-class 256x512 bf16 full training iteration, batch 16 per GPU (its per-GPU workload)', deadline, live)
            out['config1_infer_b64'] = side_leg(
                ['--mode', 'infer', '--steps', '400', '--warmup', '20', '--profile-steps', '3'],
                'configs[1]: single-class 64x128 bf16 generator-only inference (edgegan.test path: E -> G1, E -> G2), batch 64, inputs '
                'resident in HBM, up to four calls in flight', deadline, live)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(dict(num_classes=args.classes, height=args.height, width=args.width),
                                                   batch_all=args.cpu_baseline_batch)
            except Exception as e:  # the baseline must never hide the GPU number
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        if world == 1 and roof and not args.no_live_traffic:
            # roofline.traffic measured HERE (the committed table meant the driver never observed traffic)
            lt = live_traffic(roof['kernel'], args)
            if lt:
                roof['traffic'] = round(lt['bytes'], 1)
                roof['traffic_read_bytes'] = round(lt['read_bytes'], 1)
                roof['traffic_write_bytes'] = round(lt['write_bytes'], 1)
                roof['traffic_source'] = ('measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) over a '
                                          '2-iteration child run with the sub-steps serialised, averaged over %d launches of this kernel; '
                                          'read side x 2 (gfx950), KiB units' % lt['launches_sampled'])
        out['bench_seconds'] = round(time.time() - t_start, 1)
        print(json.dumps(out))
    if net is not None:
        net.close()


if __name__ == '__main__':
    main()
