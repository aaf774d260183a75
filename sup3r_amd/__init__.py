"""sup3r_amd — MI355X-native compute core for NREL/sup3r's Sup3rGan hot path.

Public surface (mirrors ``sup3r.models`` / ``sup3r.pipeline`` names for the
path this package replaces):

    from sup3r_amd import Sup3rGan, ForwardPass, ForwardPassStrategy

All arithmetic runs in ``sup3r_amd/lib/libsup3r_hip.so`` (hand-written HIP
kernels for gfx950, C-ABI in include/sup3r_hip.h).  There is no CPU fallback.
"""
__version__ = '0.1.0'

from .gan import Sup3rGan  # noqa: E402,F401
from .condmom import Sup3rCondMom  # noqa: E402,F401
from .data_centric import Sup3rGanDC  # noqa: E402,F401
from .solar_cc import SolarCC  # noqa: E402,F401
from .with_obs import Sup3rGanWithObs  # noqa: E402,F401
from .forward_pass import (ChunkPathOptions, ChunkSlicer,  # noqa: E402,F401
                           ForwardPass)
from .multi_step import MultiStepGan  # noqa: E402,F401
from .batch_queue import (DeviceBatchHandler, DeviceBatchQueue,  # noqa: E402,F401
                          DsetTuple)

__all__ = ['Sup3rGan', 'Sup3rCondMom', 'Sup3rGanDC', 'SolarCC', 'Sup3rGanWithObs', 'MultiStepGan', 'ForwardPass', 'ChunkPathOptions',
           'ChunkSlicer', 'DeviceBatchQueue', 'DeviceBatchHandler', 'DsetTuple',
           '__version__']
