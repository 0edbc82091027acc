#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -s "$@" 2>&1 | grep -vE "^\{'P'|amdgpu.ids" | tail -15
