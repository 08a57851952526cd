#!/bin/bash
# gpurun with a retry ONLY when no GPU slot is free (exit code 3); any other result is returned at once.
# usage: tools/grun.sh <timeout-seconds> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
