#!/bin/bash
# A/B of library variants built into build/variants/lib*_<name>.so (e.g. one translation unit recompiled with other compiler flags): swaps the library in place for
# each run of the given probe command and restores it.  Usage (through gpurun): bash tools/probes/ab_lib_variants.sh <prefix> "<probe command>" <name> [<name> ...]
cd $GRAFT_REPO_ROOT
L=ps-signature-and-el-passo_amd/csrc/libelpasso_hip.so
P=$1; CMD=$2; shift 2
cp $L /tmp/lib_default.so
for v in default "$@" default; do
  if [ $v = default ]; then cp /tmp/lib_default.so $L; else cp build/variants/${P}_$v.so $L; fi
  echo "== $v"
  eval "$CMD"
done
cp /tmp/lib_default.so $L
