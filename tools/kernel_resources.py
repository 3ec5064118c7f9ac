"""VGPR / SGPR / scratch / LDS per kernel of libtfft.so (CPU only: llvm-readelf --notes on the embedded gfx950 code object).
usage: python tools/kernel_resources.py [libtfft.so] [name filter]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def resources(so_path):
    tmp = tempfile.mkdtemp(prefix="tfft_res_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], cwd=tmp,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f][0]
        notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, co)], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        out[name.group(1)] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size"),
                              "lds": g("group_segment_fixed_size"), "agpr": int(blk.split()[0])}
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "tensor-fft_amd", "libtfft.so")
    flt = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for k, r in sorted(resources(path).items()):
        if flt and not any(f in k for f in flt):
            continue
        d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d)
        print(f"{d[:90]:90s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:5d}")
