#!/bin/bash
# Dev tool: VGPR / AGPR / scratch / LDS of every kernel in a built library, from the code-object notes (no GPU needed).
#   scripts/kernel_resources.sh [lib.so] [name filter]
lib=${1:-drl_graph_exploration_amd/libdrlgx.so}
filt=${2:-.}
tmp=$(mktemp -d)
(cd $tmp && /opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$OLDPWD/$lib >/dev/null 2>&1)
/opt/rocm/bin/roc-obj-ls $lib 2>/dev/null | awk '{print $NF}' | while read uri; do
  /opt/rocm/bin/roc-obj-extract "$uri" -o $tmp/co >/dev/null 2>&1 || true
done
co=$(ls $tmp/co* 2>/dev/null | head -1)
if [ -z "$co" ]; then  # fall back: readelf notes of the embedded object via llvm-objdump
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat $lib 2>/dev/null
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co.o 2>/dev/null
  co=$tmp/co.o
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print("%-60s vgpr %4s agpr %4s scratch %5s lds %7s sgpr %4s" % (name[:60], g("vgpr_count"), blk.split()[0], g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_count")))
' | grep -E "$filt"
rm -rf $tmp
