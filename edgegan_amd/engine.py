"""Host-side mirror of the reference trainer's device boundary (edgegan/models/edgegan.py).

``EdgeGAN.update_model(images, z)`` / ``EdgeGAN.infer(images, classes)`` keep the argument meaning of
the reference's ``EdgeGAN.update_model`` (models/edgegan.py:126-130) and of the two ``sess.run`` calls
of ``EdgeGAN.test`` (models/edgegan.py:606-609); the arithmetic runs in libedgegan_hip.so.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Optional

import numpy as np

from ._lib import Config, EdgeGANLibError, _CEvalOutputs, fptr, iptr, load_library, variable_specs

SUBSTEPS = ['d_optim', 'd_optim_patch2', 'd_optim_patch3', 'd_optim2', 'g_optim_u', 'e_optim', 'g_optim_b']
LOSS_NAMES = ['joint_wgan', 'joint_gp', 'image_wgan', 'image_gp', 'edge_wgan', 'edge_gp', 'loss_d_ac',
              'joint_dis_gloss', 'edge_dis_gloss', 'image_dis_gloss', 'loss_g_ac', 'zl_loss']


class EdgeGAN:
    def __init__(self, config: Config, variables: Optional[Dict[str, np.ndarray]] = None):
        self.config = config
        self._lib = load_library()
        self._ctx = C.c_void_p()
        cc = config.to_c()
        rc = self._lib.eg_create(C.byref(cc), C.byref(self._ctx))
        if rc != 0:
            raise EdgeGANLibError(f'eg_create failed ({rc}): {self._lib.eg_last_error(None).decode()}')
        self.specs = variable_specs(config)
        # alpha (random_blend, edgegan.py:33-34) is drawn per sample: under data parallelism every rank draws ITS OWN
        # rows (comm_init re-seeds by rank); eps (encoder.py:78) is ONE scalar per run: the same on every rank
        self._rng_alpha = np.random.RandomState(6666)
        self._rng_eps = np.random.RandomState(6667)
        self.rank, self.nranks = 0, 1
        if variables is not None:
            self.set_variables(variables)

    # -- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise EdgeGANLibError(f'libedgegan_hip error {rc}: {self._lib.eg_last_error(self._ctx).decode()}')

    def close(self):
        if self._ctx:
            self._lib.eg_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- variables (checkpoint layout: TF names, incl. '<var>/RMSProp' and '<var>/RMSProp_1') -----
    def set_variable(self, name: str, value: np.ndarray):
        a = np.ascontiguousarray(value, dtype=np.float32)
        self._check(self._lib.eg_set_variable(self._ctx, name.encode(), fptr(a.reshape(-1)), a.size))

    def get_variable(self, name: str, shape=None) -> np.ndarray:
        if shape is None:
            base = name
            for suf in ('/RMSProp_1', '/RMSProp', '/grad'):
                if name.endswith(suf):
                    base = name[:-len(suf)]
            shape = next(s.shape for s in self.specs if s.name == base)
        out = np.empty(int(np.prod(shape, dtype=np.int64)) if len(shape) else 1, dtype=np.float32)
        self._check(self._lib.eg_get_variable(self._ctx, name.encode(), fptr(out), out.size))
        return out.reshape(shape)

    def set_variables(self, values: Dict[str, np.ndarray]):
        for k, v in values.items():
            self.set_variable(k, v)

    def get_variables(self, slots: bool = False) -> Dict[str, np.ndarray]:
        out = {}
        for s in self.specs:
            out[s.name] = self.get_variable(s.name, s.shape)
            if slots and s.trainable:
                out[s.name + '/RMSProp'] = self.get_variable(s.name + '/RMSProp', s.shape)
                out[s.name + '/RMSProp_1'] = self.get_variable(s.name + '/RMSProp_1', s.shape)
        return out

    # -- training ---------------------------------------------------------------------------------
    def _feed(self, images, z, alpha, eps):
        cfg = self.config
        images = np.ascontiguousarray(images, dtype=np.float32)
        z = np.ascontiguousarray(z, dtype=np.float32)
        if images.shape != (cfg.batch_size, cfg.height, cfg.width, cfg.c_dim):
            raise ValueError(f'images must be {(cfg.batch_size, cfg.height, cfg.width, cfg.c_dim)}, got {images.shape}')
        if z.shape != (cfg.batch_size, cfg.z_cols):
            raise ValueError(f'z must be {(cfg.batch_size, cfg.z_cols)}, got {z.shape}')
        if alpha is None:   # the reference draws these inside the graph (edgegan.py:33-34)
            alpha = self._rng_alpha.uniform(0, 1, size=(3, cfg.batch_size))
        if eps is None:     # encoder.py:78
            eps = self._rng_eps.normal(0, 1, size=(4,))
        alpha = np.ascontiguousarray(alpha, dtype=np.float32)
        eps = np.ascontiguousarray(np.asarray(eps, dtype=np.float32).reshape(-1))
        assert alpha.shape == (3, cfg.batch_size) and eps.size >= 1
        return images, z, alpha, eps

    def upload_batch(self, images: np.ndarray, z: np.ndarray, alpha: Optional[np.ndarray] = None,
                     eps: Optional[np.ndarray] = None):
        images, z, alpha, eps = self._feed(images, z, alpha, eps)
        self._check(self._lib.eg_upload_batch(self._ctx, fptr(images), fptr(z), fptr(alpha), fptr(eps)))

    def update_model_async(self, images: np.ndarray, z: np.ndarray, alpha=None, eps=None):
        """update_model without any wait for the GPU: the feed goes through pinned host staging and is enqueued behind
        the previous iteration (eg_train_batch).  Pair with losses_async(back=1) to report iteration k-1 while k runs."""
        images, z, alpha, eps = self._feed(images, z, alpha, eps)
        self._check(self._lib.eg_train_batch(self._ctx, fptr(images), fptr(z), fptr(alpha), fptr(eps)))

    def losses_async(self, back: int = 0) -> Dict[str, float]:
        """Losses of the latest (back=0) / the previous (back=1) update_model_async call; waits for that iteration only."""
        out = np.zeros(12, dtype=np.float32)
        self._check(self._lib.eg_losses_wait(self._ctx, int(back), fptr(out)))
        return self._compose_losses(out)

    def substep(self, k: int, apply: bool = True):
        self._check(self._lib.eg_substep(self._ctx, int(k), 1 if apply else 0))

    def train_step(self):
        """One iteration on the resident batch: the 7 sub-steps of edgegan.py:109-130 (asynchronous)."""
        self._check(self._lib.eg_train_step(self._ctx))

    def update_model(self, images: np.ndarray, z: np.ndarray, alpha=None, eps=None):
        """Reference signature (models/edgegan.py:126): feed a batch and run every registered sub-step."""
        self.upload_batch(images, z, alpha, eps)
        self.train_step()

    def evaluate(self, alpha=None, eps=None, bn_updates: int = 0, tensors: bool = False) -> Dict[str, object]:
        """Post-update evaluation with the reference's semantics (edgegan.py:427-433,464-480): forward-only on the
        resident batch with the CURRENT weights and fresh alpha / eps draws.  Returns the losses (same keys as
        losses()); with tensors=True also 'd_logits' [6,B] and the generator outputs / resized tensors the reference's
        histogram and image summaries show.  bn_updates: BatchNorm moving-average updates applied per generator
        (6 mirrors the reference's seven observe-only runs per iteration; 0 = none)."""
        cfg = self.config
        B = cfg.batch_size
        alpha = self._rng_alpha.uniform(0, 1, size=(3, B)) if alpha is None else alpha
        eps = float(self._rng_eps.normal()) if eps is None else float(np.asarray(eps).reshape(-1)[0])
        alpha = np.ascontiguousarray(alpha, dtype=np.float32)
        assert alpha.shape == (3, B)
        out = {'losses12': np.zeros(12, dtype=np.float32)}
        if tensors:
            half = (B, cfg.height, cfg.width // 2, cfg.c_dim)
            def rs(size):    # (a resize to the tensor's own size is the identity)
                return half if (cfg.height, cfg.width // 2) == (size, size) else (B, size, size, cfg.c_dim)
            out.update(d_logits=np.zeros((6, B), np.float32), edge_output=np.zeros(half, np.float32),
                       image_output=np.zeros(half, np.float32))
            if cfg.use_image_discriminator:
                out.update(resized_inputs=np.zeros(rs(cfg.image_dis), np.float32), resized_image_output=np.zeros(rs(cfg.image_dis), np.float32))
            if cfg.use_edge_discriminator:
                out.update(resized_edges=np.zeros(rs(cfg.edge_dis), np.float32), resized_edge_output=np.zeros(rs(cfg.edge_dis), np.float32))
        co = _CEvalOutputs(**{k: fptr(v.reshape(-1)) for k, v in out.items()})
        self._check(self._lib.eg_eval_losses(self._ctx, fptr(alpha), eps, int(bn_updates), C.byref(co)))
        d = self._compose_losses(out.pop('losses12'))
        d.update(out)
        return d

    def losses(self) -> Dict[str, float]:
        """Losses captured INSIDE the sub-steps of the last iteration (pre-update / mixed-stage weights; free).  The
        reference prints post-update re-evaluations instead: evaluate()."""
        out = np.zeros(12, dtype=np.float32)
        self._check(self._lib.eg_read_losses(self._ctx, fptr(out)))
        return self._compose_losses(out)

    def _compose_losses(self, out) -> Dict[str, float]:
        d = dict(zip(LOSS_NAMES, out.tolist()))
        # composites the reference evaluates / prints (edgegan.py:278-321,468-486)
        d['joint_dis_dloss'] = d['joint_wgan'] + d['joint_gp']
        d['image_dis_dloss'] = d['image_wgan'] + d['image_gp']
        d['edge_dis_dloss'] = d['edge_wgan'] + d['edge_gp']
        c = self.config
        d['edge_gloss'] = c.joint_dweight * d['joint_dis_gloss'] + c.edge_dweight * d['edge_dis_gloss']
        d['image_gloss'] = (c.joint_dweight * d['joint_dis_gloss'] + c.image_dweight * d['image_dis_gloss']
                            + (d['loss_g_ac'] if c.multiclasses else 0.0))
        return d

    def sync(self):
        self._check(self._lib.eg_sync(self._ctx))

    # -- inference --------------------------------------------------------------------------------
    def infer(self, images: np.ndarray, classes: Optional[np.ndarray] = None, eps2=None):
        """E(left half) -> z (+ one-hot) -> (edge_output, image_output)  (edgegan.py:492-517,606-609)."""
        cfg = self.config
        images = np.ascontiguousarray(images, dtype=np.float32)
        nb = images.shape[0]
        assert images.shape[1:] == (cfg.height, cfg.width, cfg.c_dim)
        if eps2 is None:
            eps2 = self._rng_eps.normal(0, 1, size=(2,))
        eps2 = np.ascontiguousarray(np.asarray(eps2, dtype=np.float32).reshape(2))
        cls = None
        if cfg.multiclasses:
            if classes is None:
                raise ValueError('classes required with num_classes > 0')
            cls = np.ascontiguousarray(classes, dtype=np.int32)
        edge = np.empty((nb, cfg.height, cfg.width // 2, cfg.c_dim), dtype=np.float32)
        image = np.empty_like(edge)
        self._check(self._lib.eg_infer(self._ctx, fptr(images), iptr(cls) if cls is not None else None, nb,
                                       fptr(eps2), fptr(edge), fptr(image)))
        return edge, image

    def infer_upload(self, images: np.ndarray, classes: Optional[np.ndarray] = None, eps2=(0.0, 0.0)):
        """Makes an inference batch resident in HBM (the feed_dict copy of edgegan.py:606-609)."""
        cfg = self.config
        images = np.ascontiguousarray(images, dtype=np.float32)
        assert images.shape[1:] == (cfg.height, cfg.width, cfg.c_dim)
        cls = np.ascontiguousarray(classes, dtype=np.int32) if cfg.multiclasses else None
        eps2 = np.ascontiguousarray(np.asarray(eps2, dtype=np.float32).reshape(2))
        self._infer_nb = images.shape[0]
        self._check(self._lib.eg_infer_upload(self._ctx, fptr(images), iptr(cls) if cls is not None else None,
                                              images.shape[0], fptr(eps2)))

    def infer_run(self):
        """E -> G1, E -> G2 on the resident batch (enqueue only)."""
        self._check(self._lib.eg_infer_run(self._ctx))

    def infer_download(self):
        cfg = self.config
        edge = np.empty((self._infer_nb, cfg.height, cfg.width // 2, cfg.c_dim), dtype=np.float32)
        image = np.empty_like(edge)
        self._check(self._lib.eg_infer_download(self._ctx, fptr(edge), fptr(image)))
        return edge, image

    # -- data parallel ----------------------------------------------------------------------------
    @staticmethod
    def _locate_rccl():
        """The library binds RCCL with dlopen("librccl.so.1") (csrc/comm.cpp).  Where the loader's search path has none
        -- a host without /opt/rocm/lib on it -- point EG_RCCL_LIB at the copy PyTorch ships, found WITHOUT importing
        torch (importing it would map a second HIP runtime into a process that does not otherwise need one)."""
        if os.environ.get('EG_RCCL_LIB'):
            return
        import ctypes.util
        import importlib.util
        if ctypes.util.find_library('rccl') or any(os.path.exists(os.path.join(d, 'librccl.so.1')) for d in
                                                   ['/opt/rocm/lib'] + os.environ.get('LD_LIBRARY_PATH', '').split(':') if d):
            return
        spec = importlib.util.find_spec('torch')
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'librccl.so')
            if os.path.exists(cand):
                os.environ['EG_RCCL_LIB'] = cand

    def comm_unique_id(self) -> bytes:
        self._locate_rccl()
        buf = C.create_string_buffer(128)
        self._check(self._lib.eg_comm_unique_id(self._ctx, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        assert len(unique_id) == 128
        self._locate_rccl()
        self._check(self._lib.eg_comm_init(self._ctx, unique_id, rank, nranks))
        self.rank, self.nranks = rank, nranks
        self._rng_alpha = np.random.RandomState(6666 + 7919 * rank)

    def comm_reduce_scalars(self, values, op: str = 'sum') -> np.ndarray:
        """All-reduce of up to 64 host floats over the ranks (RCCL); returns when every rank has arrived."""
        a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
        self._check(self._lib.eg_comm_reduce_scalars(self._ctx, fptr(a), a.size, {'sum': 0, 'max': 2, 'min': 3}[op]))
        return a

    def comm_info(self) -> dict:
        """Ranks, RCCL version / library path and the gradient all-reduces enqueued so far (count, bytes)."""
        nr, ver, calls, nbytes = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
        path = C.create_string_buffer(1024)
        self._check(self._lib.eg_comm_info(self._ctx, C.byref(nr), C.byref(ver), path, len(path), C.byref(calls), C.byref(nbytes)))
        v = ver.value
        vs = None if v == 0 else (f'{v // 10000}.{v % 10000 // 100}.{v % 100}' if v >= 10000 else f'{v // 1000}.{v % 1000 // 100}.{v % 100}')
        return {'nranks': nr.value, 'rccl_version': vs, 'rccl_library': path.value.decode() or None,
                'allreduce_calls': calls.value, 'allreduce_bytes': nbytes.value}

    GROUP_NAMES = {0: 'G1+G2', 1: 'G2', 2: 'D2', 3: 'D', 4: 'D_patch2', 5: 'D_patch3', 6: 'E'}

    def comm_timing(self, enable: bool = True):
        """Event pairs around every gradient all-reduce from now on (include/edgegan_hip.h: eg_comm_timing)."""
        self._check(self._lib.eg_comm_timing(self._ctx, 1 if enable else 0))

    def comm_timing_read(self, cap: int = 4096):
        """[(group name, payload bytes, milliseconds)] of the collectives since the last read, in issue order."""
        ms = np.zeros(cap, np.float32)
        grp = np.zeros(cap, np.int32)
        nb = (C.c_longlong * cap)()
        n = C.c_int()
        self._check(self._lib.eg_comm_timing_read(self._ctx, fptr(ms), iptr(grp), nb, cap, C.byref(n)))
        return [(self.GROUP_NAMES.get(int(grp[i]), str(int(grp[i]))), int(nb[i]), float(ms[i])) for i in range(n.value)]

    def comm_barrier(self):
        if self.nranks > 1:
            self.comm_reduce_scalars([0.0])

    # -- measurement ------------------------------------------------------------------------------
    def profile_begin(self):
        self._check(self._lib.eg_profile_begin(self._ctx))

    def profile_end(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        self._check(self._lib.eg_profile_end(self._ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    def profile_layers(self) -> str:
        """CSV, one line per GEMM-shaped launch since profile_begin (label, kernel, shape, us, gflop)."""
        buf = C.create_string_buffer(8 << 20)
        self._check(self._lib.eg_profile_layers(self._ctx, buf, len(buf)))
        return buf.value.decode()

    def memory_stats(self):
        a, b = C.c_longlong(), C.c_longlong()
        self._check(self._lib.eg_memory_stats(self._ctx, C.byref(a), C.byref(b)))
        return {'arena_peak_bytes': a.value, 'arena_reserved_bytes': b.value}

    def mfma_probe(self, iters: int = 20000, random_operands: bool = True) -> float:
        """TFLOP/s of a memory-free bf16 MFMA loop on this device (include/edgegan_hip.h: eg_mfma_probe)."""
        t = C.c_double()
        self._check(self._lib.eg_mfma_probe(self._ctx, int(iters), int(bool(random_operands)), C.byref(t)))
        return t.value

    # -- test entry points ------------------------------------------------------------------------
    def test_op(self, op: str, inputs, iparams, out_shapes):
        fp = C.POINTER(C.c_float)
        ins = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) if a is not None else None for a in inputs]
        in_ptrs = (fp * len(ins))(*[fptr(a) if a is not None else None for a in ins])
        in_n = (C.c_longlong * len(ins))(*[a.size if a is not None else 0 for a in ins])
        outs = [np.zeros(int(np.prod(s, dtype=np.int64)), dtype=np.float32) if s is not None else None
                for s in out_shapes]
        out_ptrs = (fp * len(outs))(*[fptr(a) if a is not None else None for a in outs])
        out_n = (C.c_longlong * len(outs))(*[a.size if a is not None else 0 for a in outs])
        ip = np.asarray(iparams, dtype=np.int32)
        self._check(self._lib.eg_test_op(self._ctx, op.encode(), len(ins), in_ptrs, in_n, ip.size, iptr(ip),
                                         len(outs), out_ptrs, out_n))
        return [o.reshape(s) if o is not None else None for o, s in zip(outs, out_shapes)]

    def net_forward(self, net: str, x: np.ndarray, out_shape, eps: float = 0.0) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim == 2:
            x = x.reshape(x.shape[0], 1, 1, x.shape[1])
        out = np.empty(int(np.prod(out_shape)), dtype=np.float32)
        self._check(self._lib.eg_net_forward(self._ctx, net.encode(), fptr(x), x.shape[0], x.shape[1], x.shape[2],
                                             x.shape[3], float(eps), fptr(out), out.size))
        return out.reshape(out_shape)
