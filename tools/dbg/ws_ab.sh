#!/bin/bash
# same-box A/B of two library builds on the ws conv probe (tools/ab/*.so are git-ignored builds)
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  SUP3R_AMD_LIB=$PWD/$v timeout 300 python tools/dbg/ws_scaling.py MFMA_DBG=0 2>&1 | grep -v amdgpu.ids | grep None
done; done
