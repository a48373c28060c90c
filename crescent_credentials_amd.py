"""Import shim: the package directory is named `crescent-credentials_amd` (not a valid Python
identifier), so `import crescent_credentials_amd` resolves to it through this module's __path__."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "crescent-credentials_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
