#!/bin/bash
# The round-3 tree next to this one, for the same-box A/B scripts (scripts/r04_ab_*.sh run the two libraries alternately inside
# ONE gpurun call: boxes differ by +-3 %, more than most of what a round changes).  Run in the build container; build/ is
# git-ignored but travels to the GPU box with the snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$R/build/r03_tree"; mkdir -p "$R/build/r03_tree"
git -C "$R" archive 9c1b296 | tar -x -C "$R/build/r03_tree"
cd "$R/build/r03_tree" && python -c "import __graft_entry__ as g; g.build()" | tail -1
# the micro-benchmarks added after round 3 (same ABI for what they call)
cp "$R/scripts/microbench_wmse.py" "$R/scripts/microbench_step.py" "$R/build/r03_tree/scripts/" 2>/dev/null || true
