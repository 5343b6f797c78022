#!/bin/bash
# Dev tool: build a kernel-variant library for timing experiments (never shipped):  scripts/build_variant.sh <name> "<-D flags>"
# -> drl_graph_exploration_amd/libdrlgx_<name>.so, used through DRLGX_LIB_DEV=<path> (scripts/phase_profile*.py)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../drl_graph_exploration_amd/csrc"
mkdir -p _obj_$name
# only the unity build k_step.hip holds the belief kernels: every other object is shared with the product build
for f in _obj/*.o; do b=$(basename $f); [ "$b" = "k_step.o" ] || cp -u $f _obj_$name/; done
make -s OBJDIR=_obj_$name OUT=../libdrlgx_$name.so EXTRA="$flags"
ls -la ../libdrlgx_$name.so
