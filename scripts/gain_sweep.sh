#!/bin/bash
# usage: gain_sweep.sh name... : time the gain kernel with each build_alt/libcbet_<name>.so ("base" = the shipped library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/gain
for n in "$@"; do
  if [ "$n" = base ]; then unset CBET_LIB_PATH; else export CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so; fi
  echo "== $n"
  timeout -k 10 200 python3 scripts/gain_variants.py 256 60 $EXTRA 2> gpurun_out/gain/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/gain/$n.err; }
  EXTRA=
done
