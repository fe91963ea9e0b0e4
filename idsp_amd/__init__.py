"""idsp_amd — MI355X (gfx950) bulk engine for the per-sample filter hot path of
quartiq/idsp: `iir::Biquad` variants over many lanes, `hbf` half-band
decimator/interpolator cascades and the `cossin`/`Accu` DDS + `Lockin` mixer.

The product is the C-ABI shared library ``idsp_amd/lib/libidsp_hip.so``
(declared in ``include/idsp_hip.h``, sources in ``idsp_amd/csrc``);
``idsp_amd.process`` is the host-side mirror of the reference's
``dsp_process`` interface on top of it.  There is no CPU fallback.
"""
from .process import *  # noqa: F401,F403
from .process import __all__ as _process_all
from . import coefficients  # noqa: F401  (Filter, pid Builder/Pid, BiquadConfig front-end)
from .sharding import lane_shard  # noqa: F401

__all__ = list(_process_all) + ["lane_shard", "coefficients"]
