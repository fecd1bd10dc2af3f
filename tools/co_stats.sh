#!/bin/bash
# dev helper: register / scratch / spill counts of the kernels in a binary or shared library:  tools/co_stats.sh <file> [name filter]
pat=${2:-.}
d=$(mktemp -d); cp "$1" $d/bin; cd $d
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading bin > /dev/null 2>&1
for co in bin.*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | awk '
    /\.name:/ {name=$NF} /\.vgpr_count:/ {v=$NF} /\.private_segment_fixed_size:/ {p=$NF} /\.vgpr_spill_count:/ {vs=$NF} /\.sgpr_spill_count:/ {ss=$NF}
    /\.wavefront_size:/ {print name, "vgpr", v, "scratch", p, "vspill", vs, "sspill", ss}' | grep -E "$pat"
done
rm -rf $d
