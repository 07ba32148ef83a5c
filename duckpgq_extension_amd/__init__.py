"""Import shim: the product package lives in `duckpgq-extension_amd/` (the directory name the project brief
fixes, which is not a valid Python identifier).  This module makes it importable as `duckpgq_extension_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "duckpgq-extension_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
