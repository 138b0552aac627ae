"""MI355X-native DualQ-SELD-TCN hot path (gfx950 HIP kernels behind include/seld_hip.h).

The directory name is not a Python identifier; import it with
``importlib.import_module("sound-event-localization-and-detection_amd")`` or through the
``seld_amd`` alias module at the repository root.
"""
from . import _lib  # noqa: F401
from . import hip_ops  # noqa: F401
from . import hip_nn, model, train, dp, utility_functions  # noqa: F401,E402
