"""Memo of ORACLE results for the slowest GPU parity tests (tests/golden/oracle_cache/*.pt).

The oracle's CPU forward is most of those tests' wall time (a 6-window, 2-step sampler run of the CPU UNet takes 1.5 minutes
even on the GPU box's 128 cores).  An entry holds what `fn()` returned together with a key = SHA-256 over the sources it depends on
(oracle/*.py, the synthetic-weight generator, the test-case tables) and the entry's parameters: if any of them changes the key no
longer matches and the test simply recomputes the oracle, so a stale entry can never be compared against.  Entries are written by
`python tools/gen_oracle_cache.py` (CPU only; it calls the very functions the tests call) and committed; nothing under
mmgt_amd/ reads them."""
import glob
import hashlib
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE_DIR = os.path.join(ROOT, "tests", "golden", "oracle_cache")
_SOURCES = sorted(glob.glob(os.path.join(ROOT, "oracle", "*.py"))) + [
    os.path.join(ROOT, "mmgt_amd", "synthetic.py"), os.path.join(ROOT, "mmgt_amd", "unet3d_spec.py"),
    os.path.join(ROOT, "mmgt_amd", "context.py"), os.path.join(ROOT, "tests", "golden_cases.py"),
    os.path.join(ROOT, "tests", "smga_cases.py")]
_base = None


def _key(name, extra):
    global _base
    if _base is None:
        h = hashlib.sha256()
        for f in _SOURCES:
            h.update(os.path.relpath(f, ROOT).encode())
            h.update(open(f, "rb").read())
        _base = h.hexdigest()
    return hashlib.sha256((_base + "|" + name + "|" + extra).encode()).hexdigest()


def cached(name, fn, extra=""):
    """fn() -> tensors (or nested lists / dicts of tensors); served from the committed entry when its key matches."""
    path = os.path.join(CACHE_DIR, name + ".pt")
    key = _key(name, extra)
    if os.path.exists(path):
        d = torch.load(path, map_location="cpu")
        if d.get("key") == key:
            return d["value"]
    value = fn()
    if os.environ.get("MMGT_WRITE_ORACLE_CACHE") == "1":
        os.makedirs(CACHE_DIR, exist_ok=True)
        torch.save({"key": key, "value": value}, path)
    return value
