#!/bin/bash
# Dev tool: build a kernel-variant library for timing experiments (never shipped):  scripts/build_variant.sh <name> "<-D flags>" [object = k_step]
# -> drl_graph_exploration_amd/libdrlgx_<name>.so, used through DRLGX_LIB_DEV=<path> (scripts/phase_profile*.py)
set -e
name=$1; flags=$2; obj=${3:-k_step}
cd "$(dirname "$0")/../drl_graph_exploration_amd/csrc"
mkdir -p _obj_$name
# only <object>.hip is recompiled with the flags (k_step.hip, the unity build, holds the belief kernels): every other object
# is shared with the product build
for f in _obj/*.o; do b=$(basename $f); [ "$b" = "$obj.o" ] || cp -u $f _obj_$name/; done
make -s OBJDIR=_obj_$name OUT=../libdrlgx_$name.so EXTRA="$flags"
ls -la ../libdrlgx_$name.so
