#!/bin/bash
# Blocking-frame time of the two-level configurations under packet masks / library variants / packet work distribution (gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in ${CFGS:-C3 C4}; do
  for lib in ${LIBS:-libxrt.so}; do
    for pk in ${PKS:-0 7 23}; do
      XRT_LIB_VARIANT=$lib XRT_PACKET=$pk python tools/blocking.py $cfg 20 2>&1 | grep blocking | sed "s/^/$lib pk=$pk /"
    done
  done
done
