"""ctypes binding of include/edgegan_hip.h."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EG_LIB_PATH') or os.path.join(_HERE, 'libedgegan_hip.so')   # override: A/B builds


class EdgeGANLibError(RuntimeError):
    pass


class _CConfig(C.Structure):
    # include/edgegan_hip.h: eg_config.  struct_size = sizeof(eg_config) as THIS binding lays it out; the library rejects
    # any other value, so a stale field list here fails at the first call instead of reading shifted fields
    _fields_ = [('struct_size', C.c_uint),
                ('batch_size', C.c_int), ('height', C.c_int), ('width', C.c_int), ('num_classes', C.c_int),
                ('z_dim', C.c_int), ('gf_dim', C.c_int), ('df_dim', C.c_int), ('c_dim', C.c_int),
                ('use_image_discriminator', C.c_int), ('image_dis_size', C.c_int),
                ('use_edge_discriminator', C.c_int), ('edge_dis_size', C.c_int),
                ('lambda_gp', C.c_float), ('zl_weight', C.c_float),
                ('learning_rate', C.c_float), ('joint_dweight', C.c_float), ('image_dweight', C.c_float),
                ('edge_dweight', C.c_float), ('device', C.c_int), ('compute_dtype', C.c_int),
                ('e_norm', C.c_int), ('g_norm', C.c_int), ('d_norm', C.c_int)]


@dataclass
class Config:
    """Flags of the reference's edgegan/train.py:14-74 that shape the graph."""
    batch_size: int = 64
    height: int = 64
    width: int = 128
    num_classes: int = 14      # 0 == --nomulticlasses
    z_dim: int = 100
    gf_dim: int = 64
    df_dim: int = 64
    c_dim: int = 3
    dis_size: int = 128        # image_dis_size and edge_dis_size (train.py:63-67) when those are not given separately
    lambda_gp: float = 10.0
    zl_weight: float = 10.0
    learning_rate: float = 2e-4
    joint_dweight: float = 1.0
    image_dweight: float = 1.0
    edge_dweight: float = 1.0
    device: int = 0
    compute_dtype: str = 'f32'  # 'bf16': dense contractions on the bf16 MFMA (fp32 accumulate / storage)
    use_image_discriminator: bool = True   # train.py:61-62
    use_edge_discriminator: bool = True    # train.py:65-66
    image_dis_size: Optional[int] = None   # train.py:63-64 (None: dis_size)
    edge_dis_size: Optional[int] = None    # train.py:67    (None: dis_size)
    E_norm: Optional[str] = 'instance'   # train.py:54-59: 'instance' | 'batch' | None (nn/modules/normalization.py:10-29)
    G_norm: Optional[str] = 'instance'
    D_norm: Optional[str] = 'instance'

    @property
    def image_dis(self) -> int:
        return self.dis_size if self.image_dis_size is None else self.image_dis_size

    @property
    def edge_dis(self) -> int:
        return self.dis_size if self.edge_dis_size is None else self.edge_dis_size

    @property
    def multiclasses(self) -> bool:
        return self.num_classes > 0

    @property
    def z_cols(self) -> int:
        return self.z_dim + (1 if self.multiclasses else 0)

    def to_c(self) -> _CConfig:
        return _CConfig(C.sizeof(_CConfig), self.batch_size, self.height, self.width, self.num_classes, self.z_dim, self.gf_dim,
                        self.df_dim, self.c_dim, int(self.use_image_discriminator), self.image_dis,
                        int(self.use_edge_discriminator), self.edge_dis, self.lambda_gp, self.zl_weight,
                        self.learning_rate, self.joint_dweight, self.image_dweight, self.edge_dweight, self.device,
                        {'f32': 0, 'bf16': 1, 'f32x3': 2}[_dtype_override(self.compute_dtype)],
                        _NORMS[self.E_norm], _NORMS[self.G_norm], _NORMS[self.D_norm])


_NORMS = {'instance': 0, 'batch': 1, None: 2}   # (None: the identity branch, normalization.py:26-27)


def _dtype_override(dtype: str) -> str:
    """EG_F32_MODE=f32x3 | f32 re-maps the 'f32' compute type (test / A-B switch: the whole fp32 test-suite can be run on
    either fp32 implementation); explicit 'bf16' / 'f32x3' requests are never re-mapped."""
    import os
    if dtype == 'f32':
        return os.environ.get('EG_F32_MODE', 'f32')
    return dtype


class _CEvalOutputs(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ('losses12', 'd_logits', 'edge_output', 'image_output', 'resized_inputs',
                                                    'resized_image_output', 'resized_edges', 'resized_edge_output')]


EXPORTS = ['eg_default_config', 'eg_variable_count', 'eg_variable_info', 'eg_create', 'eg_destroy',
           'eg_last_error', 'eg_set_variable', 'eg_get_variable', 'eg_upload_batch', 'eg_substep',
           'eg_train_step', 'eg_train_batch', 'eg_losses_wait', 'eg_read_losses', 'eg_eval_losses', 'eg_infer', 'eg_infer_upload', 'eg_infer_run', 'eg_infer_download', 'eg_sync', 'eg_comm_unique_id', 'eg_comm_init',
           'eg_comm_reduce_scalars', 'eg_comm_info', 'eg_comm_timing', 'eg_comm_timing_read',
           'eg_profile_begin', 'eg_profile_end', 'eg_profile_layers', 'eg_memory_stats', 'eg_mfma_probe', 'eg_test_op', 'eg_net_forward', 'eg_crc32c']

_lib = None


def load_library() -> C.CDLL:
    """Loads libedgegan_hip.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdgeGANLibError(f'{LIB_PATH} is missing: build it with `make -C edgegan_amd/csrc` '
                              '(or __graft_entry__.build()); there is no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.eg_last_error.restype = C.c_char_p
    lib.eg_last_error.argtypes = [C.c_void_p]
    lib.eg_variable_count.argtypes = [C.POINTER(_CConfig)]
    lib.eg_variable_info.argtypes = [C.POINTER(_CConfig), C.c_int, C.c_char_p, C.c_int, ip, ip, ip, ip, fp]
    lib.eg_create.argtypes = [C.POINTER(_CConfig), C.POINTER(C.c_void_p)]
    lib.eg_destroy.argtypes = [C.c_void_p]
    lib.eg_set_variable.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_longlong]
    lib.eg_get_variable.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_longlong]
    lib.eg_upload_batch.argtypes = [C.c_void_p, fp, fp, fp, fp]
    lib.eg_substep.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.eg_train_step.argtypes = [C.c_void_p]
    lib.eg_read_losses.argtypes = [C.c_void_p, fp]
    lib.eg_train_batch.argtypes = [C.c_void_p, fp, fp, fp, fp]
    lib.eg_losses_wait.argtypes = [C.c_void_p, C.c_int, fp]
    lib.eg_infer.argtypes = [C.c_void_p, fp, ip, C.c_int, fp, fp, fp]
    lib.eg_eval_losses.argtypes = [C.c_void_p, fp, C.c_float, C.c_int, C.POINTER(_CEvalOutputs)]
    lib.eg_infer_upload.argtypes = [C.c_void_p, fp, ip, C.c_int, fp]
    lib.eg_infer_run.argtypes = [C.c_void_p]
    lib.eg_infer_download.argtypes = [C.c_void_p, fp, fp]
    lib.eg_sync.argtypes = [C.c_void_p]
    lib.eg_comm_unique_id.argtypes = [C.c_void_p, C.c_char_p]
    lib.eg_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.eg_comm_reduce_scalars.argtypes = [C.c_void_p, fp, C.c_int, C.c_int]
    lib.eg_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int,
                                 C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.eg_comm_timing.argtypes = [C.c_void_p, C.c_int]
    lib.eg_comm_timing_read.argtypes = [C.c_void_p, fp, ip, C.POINTER(C.c_longlong), C.c_int, ip]
    lib.eg_profile_begin.argtypes = [C.c_void_p]
    lib.eg_profile_end.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.eg_profile_layers.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
    lib.eg_memory_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.eg_mfma_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.eg_test_op.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(fp), C.POINTER(C.c_longlong), C.c_int, ip,
                               C.c_int, C.POINTER(fp), C.POINTER(C.c_longlong)]
    lib.eg_net_forward.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fp,
                                   C.c_longlong]
    lib.eg_crc32c.restype = C.c_uint
    lib.eg_crc32c.argtypes = [C.c_uint, C.c_void_p, C.c_ulonglong]
    _lib = lib
    return lib


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.POINTER(C.c_int))


@dataclass
class VarSpec:
    name: str
    shape: Tuple[int, ...]
    trainable: bool
    init_kind: int      # 0 const, 1 normal, 2 truncated normal, 3 xavier uniform
    init_param: float


def variable_specs(cfg: Config) -> List[VarSpec]:
    """The engine's variable registry (TF names / shapes of SURVEY.md Appendix C).  Needs no GPU."""
    lib = load_library()
    cc = cfg.to_c()
    n = lib.eg_variable_count(C.byref(cc))
    if n < 0:
        raise EdgeGANLibError('eg_variable_count failed')
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int * 4)()
    ndim, tr, kind = C.c_int(), C.c_int(), C.c_int()
    par = C.c_float()
    for i in range(n):
        rc = lib.eg_variable_info(C.byref(cc), i, name, 256, shape, C.byref(ndim), C.byref(tr), C.byref(kind),
                                  C.byref(par))
        if rc != 0:
            raise EdgeGANLibError(lib.eg_last_error(None).decode())
        out.append(VarSpec(name.value.decode(), tuple(shape[j] for j in range(ndim.value)), bool(tr.value),
                           kind.value, par.value))
    return out
