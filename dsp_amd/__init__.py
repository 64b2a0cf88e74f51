"""dsp_amd: MI355X-native batched echo-ranging (the per-frame chain of NUaris/dsp's airSonar_optimized.py).

Host side is Python; all arithmetic on the hot path is hand-written HIP for gfx950 behind the C ABI in
include/echo_hip.h (dsp_amd/libechohip.so, loaded with ctypes).  torch is used for device memory and
streams only.  Importing this package does not need a GPU; creating an ``EchoEngine`` does.
"""
from .config import (SonarConfig, optimized_config, original_config, realtime_config,  # noqa: F401
                     BANDS_DEFAULT, BANDS_DISTINCT, BANDS_ORIGINAL)

__all__ = ["SonarConfig", "optimized_config", "original_config", "realtime_config",
           "BANDS_DEFAULT", "BANDS_DISTINCT", "BANDS_ORIGINAL"]
