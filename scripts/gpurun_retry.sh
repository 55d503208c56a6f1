#!/bin/bash
# gpurun with a bounded retry on exit code 3 ONLY ("no box or slot free right now, nothing charged"); any other outcome is final.
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
