"""Import shim: the package directory is `fujiyama-renderer_amd/` (hyphen, per
the repo layout contract), which is not a Python identifier.  `import
fujiyama_renderer_amd` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fujiyama-renderer_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
