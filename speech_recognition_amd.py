"""Import alias: the package lives in ./speech-recognition_amd/ (a hyphen is not importable).

``import speech_recognition_amd`` executes that directory's ``__init__.py`` under this name and
points ``__path__`` at it, so ``speech_recognition_amd.models.las`` etc. resolve normally.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "speech-recognition_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
