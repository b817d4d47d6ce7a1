"""Memo of ORACLE results for the slowest GPU parity tests (tests/golden/oracle_cache/*.pt).

The oracle's CPU forward is most of those tests' wall time (a 6-window, 2-step sampler run of the CPU UNet takes 1.5 minutes
even on the GPU box's 128 cores).  An entry holds what `fn()` returned together with a key = SHA-256 over the sources it depends on
(oracle/*.py, the synthetic-weight generator, the test-case tables), the SOURCE TEXT of the functions that produce the entry
(`deps`: the oracle drivers of the test modules, their input builders, the spec functions of the side models; hashed as a
normalised syntax tree, so comments and layout do not count): if any of them changes the key no longer matches and the test simply recomputes the oracle instead of comparing against
a stale entry.  Entries are written by
`python tools/gen_oracle_cache.py` (CPU only; it calls the very functions the tests call) and committed; nothing under
mmgt_amd/ reads them."""
import ast
import glob
import hashlib
import inspect
import os
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE_DIR = os.path.join(ROOT, "tests", "golden", "oracle_cache")
# the oracle modules the cached entries are computed with (an oracle module that is no input of any entry -- wav2vec_ref.py -- is not
# listed: adding it must not invalidate them; one that becomes an input has to be added here)
_ORACLE = ("__init__", "clip_ref", "conditioning_ref", "context_ref", "ddim_ref", "pipeline_ref", "smga_ref", "unet3d_ref", "vae_ref")
_SOURCES = [os.path.join(ROOT, "oracle", n + ".py") for n in _ORACLE] + [
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


def _normalised(fn):
    """The function's abstract syntax tree without docstrings, comments or layout: a cosmetic edit does not change the key."""
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Module)) and body and \
                isinstance(body[0], ast.Expr) and isinstance(body[0].value, ast.Constant) and isinstance(body[0].value.value, str):
            node.body = body[1:] or [ast.Pass()]
    # (ast.unparse, not ast.dump: the dump's text changed in Python 3.13 -- defaulted fields are omitted -- which would silently invalidate every
    # committed entry on an interpreter bump; unparse prints source in one canonical layout -- ADVICE r4)
    return ast.unparse(tree)


def _deps_digest(deps):
    # (no torch version in the key: the entries are compared at tolerances far above kernel-to-kernel rounding differences of CPU
    # PyTorch builds, and a key that changes with the build would turn the CPU suite's currency check into a failure -- ADVICE r3)
    h = hashlib.sha256()
    for d in deps:
        h.update(getattr(d, "__qualname__", repr(d)).encode())
        h.update(_normalised(d).encode())
    return h.hexdigest()


def entry_specs():
    """name -> (extra, deps) of every entry: the parameters and the callables (outside the always-hashed sources) whose source
    text the entry depends on.  One table for the tests, tools/gen_oracle_cache.py and the currency check of the CPU suite."""
    from mmgt_amd.side_models import pose_guider_spec
    from mmgt_amd.unet3d_spec import unet2d_reference_spec
    from mmgt_amd.vae import vae_decoder_spec
    from tests import golden_cases as gc
    from tests import test_pipeline_gpu as TP
    from tests import test_smga as TS
    from tests import test_unet_gpu as TU
    from tests import test_vae as TV
    pipe = (TP._inputs, TP.build_weights, pose_guider_spec, unet2d_reference_spec, vae_decoder_spec)
    return {
        "pipeline_fp32_8_12_4": ("8,12,4", pipe + (TP.oracle_pipeline_fp32,)),
        "pipeline_fp32_14_8_2": ("14,8,2", pipe + (TP.oracle_pipeline_fp32,)),
        "pipeline_bf16_floor": ("", pipe + (TP.oracle_pipeline_bf16_floor,)),
        "long_video_96": ("", pipe + (TP.oracle_long_video,)),
        "unet_512x512_six_frames": (repr(sorted(TU.SIX_FRAME_CASE.items())), (TU._run_oracle,)),
        "smga_sampler_bf16_floor": ("", (TS.smga_bf16_floor,)),
        "unet_512x512_twelve_frames": (repr(sorted(TU.TWELVE_FRAME_CASE.items())), (TU._run_oracle,)),
        "vae_decode_512x512_frame": ("", (TV._sd, TV.oracle_full_resolution_frame, vae_decoder_spec)),
    }


def entry_key(name):
    extra, deps = entry_specs()[name]
    return _key(name, extra + "|" + _deps_digest(deps))


def cached(name, fn):
    """fn() -> tensors (or nested lists / dicts of tensors); served from the committed entry when its key matches."""
    path = os.path.join(CACHE_DIR, name + ".pt")
    key = entry_key(name)
    if os.path.exists(path):
        d = torch.load(path, map_location="cpu")
        if d.get("key") == key:
            return d["value"]
    value = fn()
    if os.environ.get("MMGT_WRITE_ORACLE_CACHE") == "1":
        os.makedirs(CACHE_DIR, exist_ok=True)
        torch.save({"key": key, "value": value}, path)
    return value
