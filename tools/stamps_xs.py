"""Cycles per phase of the row-stationary conv / Linear kernel (csrc/conv_xs.hip) over Mimi chunk decodes, as seen by thread 0
of every workgroup: a diagnostic build VARIANT with clock64() stamps (never the product library).
Run on the GPU box from the repo root:  python tools/stamps_xs.py"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib = subprocess.run([sys.executable, "-m", "smoltts_amd.build", "--variant", "stamps_xs", "--flags=-DSMOLTTS_DBG_XS_STAMPS"],
                     capture_output=True, text=True, check=True, cwd=ROOT).stdout.strip().splitlines()[-1]
os.environ["SMOLTTS_LIB"] = lib
import torch  # noqa: E402

from smoltts_amd import engine as E  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.engine import MimiEngine, MimiSession  # noqa: E402

L = E.load_library()
fn = ctypes.CDLL(lib).smoltts_debug_xs_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
fk = ctypes.CDLL(lib).smoltts_debug_ks_stamps
fk.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
B, F = 32, 32
eng = MimiEngine(synthetic_mimi_state(seed=0), 8, window=0, max_positions=2 * F * 8 + 16)
sess = MimiSession(eng, max_batch=B, max_chunk_frames=F)
codes = torch.randint(0, 2048, (B, F * 8, 8), dtype=torch.int32, device="cuda")
pcm = torch.zeros(B, F * 8 * 1920, device="cuda")
out, outk = (ctypes.c_ulonglong * 64)(), (ctypes.c_ulonglong * 64)()
for i in range(3):
    sess.decode_chunk(codes, i * F, F, pcm, code_offset=0)
torch.cuda.synchronize()
fn(out, 1)
fk(outk, 1)
n = 4
for i in range(3, 3 + n):
    sess.decode_chunk(codes, i * F, F, pcm, code_offset=0)
torch.cuda.synchronize()
fn(out, 0)
fk(outk, 0)
rows = {0: "SEANet conv, N = 640", 2: "SEANet conv, N = 128", 1: "k1 conv + residual", 5: "wqkv (LayerNorm in, RoPE out)", 4: "wo (scale + residual)",
        3: "fc1 (LayerNorm in, GELU out)", 6: "fc2 K parts (partial sums out)"}
phases = ["rows -> pieces in LDS", "first W loads + barrier", "K loop", "epilogue"]
print("cycles per workgroup (thread 0); 100 cycles = 0.042 us at 2.4 GHz")
for r, nm in rows.items():
    wg = out[r * 8 + 6]
    if not wg:
        continue
    tot = sum(out[r * 8 + i] for i in range(4))
    print(f"{nm:32s} {wg // n:6d} workgroups per chunk: " + "  ".join(f"{ph} {out[r * 8 + i] / wg:7.0f}" for i, ph in enumerate(phases)) + f"   sum {tot / wg:7.0f}")
print("conv_ks.hip (> 256 channels, 128-channel slices):")
rows = {1: "conv0 512 -> 1024 k7 (NTW 1)", 4: "ConvTranspose 1024 -> 512 (NTW 4)", 2: "res1 k3 conv 512 -> 256 (NTW 2)", 3: "ConvTranspose 512 -> 256 (NTW 4, 3 parts)"}
phases = ["rows -> pieces (all slices)", "barriers", "K loops", "epilogue"]
for r, nm in rows.items():
    wg = outk[r * 8 + 6]
    if not wg:
        continue
    tot = sum(outk[r * 8 + i] for i in range(4))
    print(f"{nm:42s} {wg // n:6d} workgroups per chunk: " + "  ".join(f"{ph} {outk[r * 8 + i] / wg:7.0f}" for i, ph in enumerate(phases)) + f"   sum {tot / wg:7.0f}")
