#!/bin/bash
# Victims that are torch's own fp32 kernels (gradient accumulation, an all-reduce's sum) beside MFMA partners of this library.
set -u
S=${1:-12}
for V in add_ sum2 copy fma; do
  for P in k_sp k_gather step; do
    echo "== victim $V, partner $P, $S s"
    python scripts/hw/atomic_share_stress.py --victim $V --partner $P --seconds $S 2>&1 | grep -E "repetitions|launches|training steps|Error|error"
  done
done
