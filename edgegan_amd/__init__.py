"""edgegan_amd -- MI355X (gfx950) engine for the EdgeGAN training / inference hot path.

Thin Python host over the C ABI of ``libedgegan_hip.so`` (include/edgegan_hip.h).  There is no
CPU fallback: every compute call requires the HIP library and a GPU and fails loudly otherwise.
"""
from ._lib import EdgeGANLibError, Config, load_library, variable_specs   # noqa: F401
from .variables import init_variables                                      # noqa: F401
from .engine import EdgeGAN                                                # noqa: F401
from .synth import synthetic_batch                                         # noqa: F401
