"""Imports the package directory `bark.cpp_amd/` (its name holds a dot, so a plain `import` cannot reach it)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "bark_cpp_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(_ROOT, "bark.cpp_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
