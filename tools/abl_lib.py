"""The timing-ablation build of the library (mmgt_amd/libmmgt_hip_abl.so, `make -C mmgt_amd/csrc abl`): the only library that contains the ABL / DBG
instantiations of tleg / gnconv / ffn / rowgemm (results are garbage by construction) and accepts the mmgt_tune keys that select them.
The instruments that time ablations call use() BEFORE importing mmgt_amd.hip; the product library refuses those keys."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use():
    path = os.path.join(ROOT, "mmgt_amd", "libmmgt_hip_abl.so")
    subprocess.run(["make", "-C", os.path.join(ROOT, "mmgt_amd", "csrc"), "-j", "8", "abl"], check=True, stdout=subprocess.DEVNULL)
    os.environ["MMGT_LIB"] = path
    return path
