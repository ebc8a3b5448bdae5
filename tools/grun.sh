#!/bin/bash
# Build container: gpurun with a retry while no GPU slot / box is free (exit code 3: nothing was charged).  usage: grun.sh LOG TIMEOUT 'command'
LOG=$1; TMO=$2; shift 2
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
