"""Build ``libsmoltts_hip.so`` (hipcc, gfx950 only) in-tree next to its sources.

``python -m smoltts_amd.build`` or ``build_library()``; ``__graft_entry__.build()`` calls this.
hipcc cross-compiles without a GPU.  The product never falls back to anything else: if the
library is missing, ``smoltts_amd.engine.load_library`` raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libsmoltts_hip.so"
SOURCES = ["api.hip", "gemm.hip", "gemm3.hip", "attention.hip", "small_ops.hip", "lm_engine.hip", "mimi_engine.hip", "mimi_encoder.hip"]
ARCH = "gfx950"


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


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    headers = [CSRC / "common.h", CSRC / "x3.h", CSRC / "mimi_common.h", CSRC.parents[1] / "include" / "smoltts_hip.h"]
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
             *os.environ.get("SMOLTTS_HIPCC_FLAGS", "").split()]  # extra -D switches for A/B experiments

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
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
