#!/bin/bash
# The round-3 tree next to this one, for the same-box A/B scripts (scripts/r04_ab_*.sh run the two libraries alternately inside
# ONE gpurun call: boxes differ by +-3 %, more than most of what a round changes).  Run in the build container; build/ is
# git-ignored but travels to the GPU box with the snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$R/build/r03_tree"; mkdir -p "$R/build/r03_tree"
git -C "$R" archive 9c1b296 | tar -x -C "$R/build/r03_tree"
cd "$R/build/r03_tree" && python -c "import __graft_entry__ as g; g.build()" | tail -1
# the micro-benchmark added after round 3 (same ABI for what it calls); the "replacec" mode of microbench_step.py (the engine's
# replace launch, table folded in) was added to the round-3 copy by hand: the phase entry and the t_ve / t_abt / t_rsig / coef_out
# fields, as in this tree's scripts/microbench_step.py
cp "$R/scripts/microbench_wmse.py" "$R/build/r03_tree/scripts/"
