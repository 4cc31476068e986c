"""Import shim: exposes the `halo2-zkcert_amd/` directory as the package `halo2_zkcert_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "halo2-zkcert_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
