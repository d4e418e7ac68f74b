"""Import shim: the package directory is named ``wav2vec-s_amd`` (not a valid Python
identifier), so ``import wav2vec_s_amd`` resolves here and forwards to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "wav2vec-s_amd")]
__package__ = "wav2vec_s_amd"
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
