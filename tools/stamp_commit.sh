#!/bin/bash
# In the build container, before a profiling run on a GPU box: record the commit (and whether the tree differs from it) in tools/.commit,
# which travels with the snapshot; the PMC summaries written on the box quote it.
cd "$(dirname "$0")/.." && { echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- grates_amd/csrc include || echo '+uncommitted changes in csrc')"; } > tools/.commit && cat tools/.commit
