#!/bin/bash
# development gpurun call without the 62 MB of golden fixtures (tests that need them must not be in the command): the push
# is charged GPU time.  Usage: scripts/devrun.sh <timeout-seconds> '<command>'.  Restores .gpurunignore afterwards.
cd /root/repo
cp .gpurunignore /tmp/.gpurunignore.keep
printf 'tests/golden/\ndynamicpdb_amd/csrc/build/\nprofiles/\n' >> .gpurunignore
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
rc=$?
cp /tmp/.gpurunignore.keep .gpurunignore
exit $rc
