"""Build ``libsmoltts_hip.so`` (hipcc, gfx950 only) in-tree next to its sources.

``python -m smoltts_amd.build`` or ``build_library()``; ``__graft_entry__.build()`` calls this.
hipcc cross-compiles without a GPU.  The product never falls back to anything else: if the
library is missing, ``smoltts_amd.engine.load_library`` raises.

The product library (``csrc/libsmoltts_hip.so``) is always compiled with exactly ``PRODUCT_FLAGS``: its objects carry a
record of the flags they were built with and are rebuilt when that record differs.  Experiments that need other flags
(``-DSMOLTTS_DEBUG_HOOKS``, the ``-DSMOLTTS_DBG_*`` A/B switches of tools/) are *variants*: they go to
``csrc/variants/<name>/`` with their own objects and are loaded explicitly (``load_library(path=...)`` or
``SMOLTTS_LIB=<path>``), so an A/B run can never leave the product library in a non-product state.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Optional, Sequence

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libsmoltts_hip.so"
SOURCES = ["api.hip", "gemm.hip", "gemm_b3.hip", "gemm3.hip", "attention.hip", "small_ops.hip", "lm_engine.hip", "mimi_engine.hip", "mimi_encoder.hip",
           "seanet.hip", "seanet_last.hip", "conv_xs.hip", "conv_ks.hip"]
ARCH = "gfx950"
# -amdgpu-kernarg-preload-count: the leading scalar arguments of a kernel arrive in SGPRs at wave launch instead of behind a
# scalar load from memory nobody has touched since the last replay (gemm3.hip, attention.hip: the frame's launches put what
# their first loads need in front of their argument struct; measured +2 % frames/s)
PRODUCT_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
NAMED_VARIANTS = {"hooks": ["-DSMOLTTS_DEBUG_HOOKS"],  # event hooks + in-kernel cycle stamps for tools/
                  "knobs": ["-DSMOLTTS_DBG_KNOBS"]}    # the experiment environment switches of tools/ (forced tiles, kernels off)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def variant_path(name: str) -> Path:
    return CSRC / "variants" / name / "libsmoltts_hip.so"


def build_library(force: bool = False, verbose: bool = False, variant: Optional[str] = None,
                  extra_flags: Optional[Sequence[str]] = None) -> Path:
    """Product library when ``variant`` is None (``extra_flags`` is then refused); otherwise the named variant with
    ``extra_flags`` (default: ``NAMED_VARIANTS[variant]``) appended to the product flags, in its own directory."""
    hipcc = _hipcc()
    if variant is None:
        if extra_flags:
            raise ValueError("extra flags build a variant: pass variant=<name> (the product library has fixed flags)")
        objdir, lib, flags = CSRC / "build", LIB, list(PRODUCT_FLAGS)
    else:
        if not variant.replace("_", "").replace("-", "").isalnum():
            raise ValueError(f"bad variant name {variant!r}")
        extra = list(extra_flags) if extra_flags is not None else list(NAMED_VARIANTS.get(variant, []))
        if not extra:
            raise ValueError(f"variant {variant!r} has no flags (known names: {sorted(NAMED_VARIANTS)})")
        lib = variant_path(variant)
        objdir, flags = lib.parent / "obj", list(PRODUCT_FLAGS) + extra
    headers = sorted(CSRC.glob("*.h")) + [CSRC.parents[1] / "include" / "smoltts_hip.h"]
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objdir.mkdir(parents=True, exist_ok=True)
    # objects are only as good as the flags that made them: a changed flag set forces recompilation
    stamp = objdir / "flags.txt"
    want = hashlib.sha256(" ".join(flags).encode()).hexdigest() + "\n" + " ".join(flags) + "\n"
    if not stamp.exists() or stamp.read_text() != want:
        force = True

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src, *headers]):
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    stamp.write_text(want)
    if force or _stale(lib, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default=None, help=f"build csrc/variants/<name>/ instead of the product library (named: {sorted(NAMED_VARIANTS)})")
    ap.add_argument("--flags", default=None, help="extra hipcc flags of the variant, e.g. '-DSMOLTTS_DBG_PIECES=2'")
    a = ap.parse_args()
    print(build_library(force=a.force, verbose=True, variant=a.variant, extra_flags=a.flags.split() if a.flags else None))
