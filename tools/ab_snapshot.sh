#!/bin/bash
# Same-box A/B: boxes differ by +-2 % from call to call, so a change is judged against a snapshot of the last accepted tree
# benched in the SAME gpurun call.  This writes that snapshot (from a git revision, default HEAD) to .ab_base/ -- git-ignored,
# shipped by gpurun -- and builds its library;  tools/gpu_check.sh ... ab  then runs  base, new, base, new.
set -e
rev=${1:-HEAD}
cd "$(dirname "$0")/.."
rm -rf .ab_base && mkdir .ab_base
git archive "$rev" pixray_amd bench.py include oracle | tar -x -C .ab_base
make -C .ab_base/pixray_amd/csrc -j 32 > /dev/null
rm -f .ab_base/pixray_amd/csrc/*.o
echo "snapshot of $rev -> .ab_base/ ($(git rev-parse --short $rev))" | tee .ab_base/REV
