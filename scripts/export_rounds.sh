#!/bin/bash
# The final trees of earlier rounds, exported and built side by side for scripts/ab_rounds.py (same-box A/B across rounds):
# build/rounds/rNN = `git archive` of the commit BEFORE "round N: VERDICT + ADVICE + BENCH" (= the tree the driver benchmarked),
# with its own librobchar_hip.so and oracle port built in place.  build/ is git-ignored and travels to the GPU box.
set -e
cd "$(dirname "$0")/.."
for n in 2 3 4; do
    c=$(git log --format=%H --grep="^round $n: VERDICT" | head -1)
    [ -n "$c" ] || { echo "no verdict commit for round $n"; exit 1; }
    d=build/rounds/r0$n
    rm -rf "$d"; mkdir -p "$d"
    git archive "${c}^" | tar -x -C "$d"
    make -C "$d/code-robchar_amd/csrc" > /dev/null
    make -C "$d/oracle" > /dev/null
    echo "r0$n = $(git rev-parse --short ${c}^): $(ls -la $d/code-robchar_amd/csrc/librobchar_hip.so | awk '{print $5}') bytes"
done
