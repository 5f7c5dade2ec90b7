#!/bin/bash
# Wrapper around gpurun: records the commit (and whether the tree is dirty) the snapshot is taken from in .git_head,
# which travels with the snapshot (the GPU box has no .git) so that profiles can name the commit they were taken at.
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet HEAD -- depthhead_amd bench.py tools || echo +dirty)" > .git_head
exec /usr/local/graft/bin/gpurun "$@"
