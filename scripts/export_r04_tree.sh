#!/bin/bash
# The round-4 tree (commit 4d1d1b1, the head the round-4 verdict judged) next to this one, for the same-box A/B scripts
# (scripts/r05_ab_*.sh run the two libraries alternately inside ONE gpurun call: boxes differ by +-3 %, more than most of what a
# round changes).  Run in the build container; build/ is git-ignored but travels to the GPU box with the snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$R/build/r04_tree"; mkdir -p "$R/build/r04_tree"
git -C "$R" archive 4d1d1b1 | tar -x -C "$R/build/r04_tree"
cd "$R/build/r04_tree" && python -c "import __graft_entry__ as g; g.build()" | tail -1
