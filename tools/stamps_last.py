"""Cycles per phase of the fused last SEANet stage (csrc/seanet_last.hip), as seen by thread 0 of every (persistent) workgroup:
a diagnostic build VARIANT with clock64() stamps at the phase boundaries (never the product library).
Run on the GPU box from the repo root:  python tools/stamps_last.py"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib = subprocess.run([sys.executable, "-m", "smoltts_amd.build", "--variant", "stamps_last", "--flags=-DSMOLTTS_DBG_LAST_STAMPS"],
                     capture_output=True, text=True, check=True, cwd=ROOT).stdout.strip().splitlines()[-1]
os.environ["SMOLTTS_LIB"] = lib
import torch  # noqa: E402

from smoltts_amd import engine as E  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.engine import MimiEngine, MimiSession  # noqa: E402

L = E.load_library()
fn = ctypes.CDLL(lib).smoltts_debug_last_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B, F = 32, 32
eng = MimiEngine(synthetic_mimi_state(seed=0), 8, window=0, max_positions=2 * F * 8 + 16)
sess = MimiSession(eng, max_batch=B, max_chunk_frames=F)
codes = torch.randint(0, 2048, (B, F * 8, 8), dtype=torch.int32, device="cuda")
pcm = torch.zeros(B, F * 8 * 1920, device="cuda")
out = (ctypes.c_ulonglong * 16)()
for i in range(3):
    sess.decode_chunk(codes, i * F, F, pcm, code_offset=0)
torch.cuda.synchronize()
fn(out, 1)
n = 4
for i in range(3, 3 + n):
    sess.decode_chunk(codes, i * F, F, pcm, code_offset=0)
torch.cuda.synchronize()
fn(out, 0)
names = ["input pieces -> LDS", "barrier", "phase A (ConvTranspose MFMAs)", "barrier", "phase 0 (half 0)", "barrier", "D + phase B (both halves)",
         "barrier", "phase C (+ next phase 0) (both halves)", "barrier"]
tiles = 244 * B
tot = sum(out[i] for i in range(10))
print(f"cycles per tile (thread 0 of 256 workgroups, {n} launches, {tiles} tiles per launch); 100 cycles = 0.042 us at 2.4 GHz")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {out[i] / (n * tiles):9.0f}  {100.0 * out[i] / tot:5.1f} %")
print(f"  {'sum':42s} {tot / (n * tiles):9.0f}")
