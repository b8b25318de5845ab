"""Import shim: the package directory is `tum-control_amd/` (not a valid Python identifier),
this module makes it importable as `tum_control_amd` (sub-modules resolve through __path__)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tum-control_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
