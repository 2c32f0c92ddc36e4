#!/bin/bash
# usage: scripts/ab_lib.sh <command...>   runs the command once per library in tmp_libs/ (A/B of two builds, same box)
set -e
for l in tmp_libs/*.so; do
  cp "$l" fcvsr_amd/lib/libfcvsr_hip.so
  echo "== $l"
  "$@"
done
