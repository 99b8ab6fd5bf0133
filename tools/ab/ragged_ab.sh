#!/bin/bash
# A/B of two builds of libpire_hip.so on the same box: tools/ragged_case.py with the current and with an older library.
# tools/ab/libpire_hip_old.so is not in the repository (*.so is git-ignored): build it from the commit to compare with
#   git worktree add /tmp/old <commit> && make -C /tmp/old/pire_amd/csrc && cp /tmp/old/pire_amd/libpire_hip.so tools/ab/libpire_hip_old.so
# (built .so files travel to the GPU box with the snapshot).
set -u
cp pire_amd/libpire_hip.so /tmp/new.so
for round in 1 2; do
for which in new old; do
  if [ $which = old ]; then cp tools/ab/libpire_hip_old.so pire_amd/libpire_hip.so; else cp /tmp/new.so pire_amd/libpire_hip.so; fi
  for c in urls loglines uniform2k fixed4096; do
    echo -n "$which $round: "; PYTHONPATH=. timeout 120 python tools/ragged_case.py $c 3 2>&1 | grep "^ragged"
  done
done
done
cp /tmp/new.so pire_amd/libpire_hip.so
