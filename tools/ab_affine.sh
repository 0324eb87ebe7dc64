#!/bin/bash
# same-box A/B of the affine warp: tools/ab_affine.sh NAME1 [NAME2 ...] (variants built by tools/build_variant.py NAME --src=affine.hip ...)
# a NAME of the form VAR=VALUE runs the stock library with that environment variable instead
set -e
for v in default "$@" default; do
  unset BHCORE_LIB
  case $v in
    default) ;;
    *=*) export "$v" ;;
    *) export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_$v.so ;;
  esac
  echo "== $v"; python tools/affine_probe.py
  case $v in *=*) unset "${v%%=*}" ;; esac
done
