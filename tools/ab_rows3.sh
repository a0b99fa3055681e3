#!/bin/bash
# A/B on one box: the codec chunk with the bf16x3 chunk attention (attn_rows3_kernel) and with the fp32 one (knobs build, SMOLTTS_ROWS3=0).
# Chunk time at stream start (tools/time_mimi.py) and at stream positions ~900 (tools/mimi_indirect_cost.py, if present).
KN=smoltts_amd/csrc/variants/knobs/libsmoltts_hip.so
for i in 1 2; do
  echo "== rows3 on"; SMOLTTS_LIB=$KN timeout -k 10 120 python tools/time_mimi.py 32 32 ${NCH:-25} 2>/dev/null | grep -i "chunk"
  echo "== rows3 off"; SMOLTTS_LIB=$KN SMOLTTS_ROWS3=0 timeout -k 10 120 python tools/time_mimi.py 32 32 ${NCH:-25} 2>/dev/null | grep -i "chunk"
done
