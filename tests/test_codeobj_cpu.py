"""The built library's gfx950 code objects: no kernel of the hot path may spill to scratch memory.

A wave that keeps its operand fragments in private memory instead of registers still computes the right numbers, so no
parity test notices -- only the clock does (round 3: a helper lambda that captured the fragment arrays put the K = 3072
GEMM's fragments into scratch and the frame took three times as long).  The kernel descriptors in the code objects say what
each kernel uses: `private_segment_fixed_size` must be 0 except for the few instantiations listed below, none of which the
benchmark's models reach."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")

# kernel-name patterns that are allowed some scratch, with the reason
ALLOWED = [
    (r"attn_split_kernelILi4E", "GQA groups of 4 query heads per kv head at 1024 threads: no shipped config (150m / 70m use 3)"),
    (r"gemm_kernelILb1ELi2ELi3ELi3ELi[35]E", "fp32 Mimi skinny GEMM, 3 x 3 tiles per wave: an encoder-side shape outside the decode loop"),
]


@pytest.mark.skipif(not (LLVM / "llvm-objdump").exists(), reason="ROCm LLVM tools not installed")
def test_no_hot_path_kernel_uses_scratch(tmp_path):
    from smoltts_amd.build import LIB, build_library

    build_library()
    so = tmp_path / "lib.so"
    shutil.copy(LIB, so)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", so.name], cwd=tmp_path, check=True, capture_output=True)  # extracts the bundles beside it
    objs = sorted(tmp_path.glob("lib.so.*gfx950"))
    assert objs, "no gfx950 code object in the library"
    seen, bad = 0, []
    for o in objs:
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(o)], check=True, capture_output=True, text=True).stdout
        name = None
        for ln in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
            if m and name:
                seen += 1
                if int(m.group(1)) > 0 and not any(re.search(pat, name) for pat, _ in ALLOWED):
                    bad.append((name, int(m.group(1))))
    assert seen > 200, f"only {seen} kernel descriptors found"
    assert not bad, f"kernels with scratch (bytes per lane): {bad}"
