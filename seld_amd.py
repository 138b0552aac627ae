"""Import alias: `import seld_amd` == the package directory `sound-event-localization-and-detection_amd`."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("sound-event-localization-and-detection_amd")
sys.modules[__name__] = _pkg
