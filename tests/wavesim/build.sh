#!/bin/bash
# TEST INFRASTRUCTURE: compiles the engine's own sources (robopianist_amd/csrc/rp_engine.hip) for the CPU
# wave emulator.  Output: tests/wavesim/_build/librp_engine_wavesim.so (git-ignored).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/_build"
mkdir -p "$out"
CXX=${CXX:-g++}
OPT=${WAVESIM_OPT:--O1}
$CXX -std=c++17 $OPT -g -fPIC -DRPK_POISON_LDS -shared -fno-strict-aliasing -Wno-unused-value -Wno-attributes \
  -I"$here" -I"$root/robopianist_amd/csrc" $WAVESIM_DEFS \
  -x c++ "$root/robopianist_amd/csrc/rp_engine.hip" -x c++ "$here/wavesim.cpp" \
  -o "$out/librp_engine_wavesim.so" -lpthread -ldl
echo "built $out/librp_engine_wavesim.so"
