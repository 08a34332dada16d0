#!/bin/bash
# One runner for GPU-box measurement sessions (round 4: replaces the 33 one-off tools/probe_r03_*.sh of round 3 -- `git show c96e3fc:tools/`
# has them; what they measured is summarised file by file in profiles/README.md).
#
#   tools/probe.sh <name> '<label> <timeout s> <command ...>' ['<label> <timeout s> <command ...>' ...]
#
# Every step runs under `timeout -k 10`, writes gpurun_out/probe_<name>/<label>.log, appends "<label> rc=<code>" to rc.txt and prints the last
# lines of its log; a step that fails or times out ends the session (no further GPU step after a kill, see the gpurun rules).  The usual
# step bodies:
#   python3 tools/rate.py prec:nx,ny,nz:solver[:precond][@SWITCH=value;...] ...      rates / ms per iteration, switches through the API
#   python3 tools/kwin_sweep.py [kwin ...]                                            k windows of the two-stage pass
#   tools/small_trace.sh                                                              kernel durations on small grids (rocprofv3)
#   tools/profile_r04.sh [bench|pmc32|pmc64|sq|all]                                   the profiles/ record of a round
#   python -m pytest tests -m gpu -x -q [-k expr]
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
name=$1; shift
O=gpurun_out/probe_$name
mkdir -p $O
: > $O/rc.txt
for step in "$@"; do
  set -- $step
  label=$1; limit=$2; shift 2
  timeout -k 10 "$limit" "$@" > $O/$label.log 2>&1
  rc=$?
  echo "$label rc=$rc" | tee -a $O/rc.txt
  tail -${PROBE_TAIL:-25} $O/$label.log
  [ $rc -eq 0 ] || exit $rc
done
