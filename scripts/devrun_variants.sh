#!/bin/bash
# development gpurun call that ships the diagnostic library variants (dynamicpdb_amd/csrc/variants, normally gpurun-ignored) and
# leaves the golden fixtures at home.  Usage: scripts/devrun_variants.sh <timeout-seconds> '<command>'
cd /root/repo
cp .gpurunignore /tmp/.gpurunignore.keep
grep -v "csrc/variants" /tmp/.gpurunignore.keep > .gpurunignore
printf 'tests/golden/\ndynamicpdb_amd/csrc/build/\nprofiles/\ndynamicpdb_amd/csrc/variants/obj_*\n' >> .gpurunignore
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
rc=$?
cp /tmp/.gpurunignore.keep .gpurunignore
exit $rc
