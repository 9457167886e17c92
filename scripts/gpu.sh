#!/bin/bash
# ONE parametrised runner for the GPU sessions (round 5 on; replaces the per-call scripts gpu_r3_*.sh / gpu_r4_*.sh):
#   scripts/gpu.sh <label> [--timeout SECONDS] -- '<command line run on the GPU box from the repo root>'
# -> /usr/local/graft/bin/gpurun with the command wrapped so that its stdout/stderr land in gpurun_out/<label>/log.txt
# (merged back here when the call ends) and the command line itself in gpurun_out/<label>/cmd.txt.  scripts/README.md keeps
# the table "profiles/<file>  <-  label + command".
set -e
label=$1; shift
timeout=900
if [ "$1" = "--timeout" ]; then timeout=$2; shift 2; fi
[ "$1" = "--" ] && shift
cmd="$*"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/$label
printf '%s\n' "$cmd" > gpurun_out/$label/cmd.txt
exec /usr/local/graft/bin/gpurun --timeout $timeout -- "mkdir -p gpurun_out/$label && export ROBCHAR_TEST_NO_BUILD=1 && ( $cmd ) > gpurun_out/$label/log.txt 2>&1; rc=\$?; tail -40 gpurun_out/$label/log.txt; exit \$rc"
