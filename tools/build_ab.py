"""Builds an experimental copy of the library for in-process A/B timing: python tools/build_ab.py NAME [-DMACRO ...]
-> build/libtfft_NAME.so (outside the package directory; same flags as the shipped build plus the given defines). tools/ab_variants.py takes
`lib=NAME:VARIANT` specs to time it next to the shipped build in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name = sys.argv[1]
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
target = os.path.join(ROOT, "build", f"libtfft_{name}.so")
g._compile_lib(target, tuple(sys.argv[2:]))
print("built", target)
