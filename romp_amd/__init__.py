"""romp_amd -- MI355X-native ROMP inference hot path (drop-in for ``simple_romp/romp``).

``import romp_amd as romp`` gives the reference's public surface (romp/__init__.py:1-2):
``ROMP, romp_settings, ResultSaver, WebcamVideoStream``.
"""
from .main import ROMP, romp_settings  # noqa: F401
from .utils import ResultSaver, WebcamVideoStream  # noqa: F401
from . import main  # noqa: F401
