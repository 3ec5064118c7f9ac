"""Import alias: `import tensor_fft_amd` loads the package in ./tensor-fft_amd/.

The package directory keeps the upstream project's hyphenated name, which Python
cannot import directly; this module loads it under an importable name and puts
it in its own place in sys.modules.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tensor-fft_amd")
_spec = importlib.util.spec_from_file_location(
    "tensor_fft_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tensor_fft_amd"] = _mod
_spec.loader.exec_module(_mod)
