"""bark.cpp_amd - MI355X-native Bark engine: HIP/C++ library (csrc/ -> lib/libbark.so) + this thin
ctypes mirror of its C API.

The directory name contains a dot, so import it through `load_package()` of the repo-root helper
`bark_amd_loader.py`, or put the directory itself on sys.path and `import api`.
"""
from .api import (BarkContext, Batcher, BarkContextParams, BarkHipStats, build_library, default_params, library_path,  # noqa: F401
                  load_library)
