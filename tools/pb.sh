#!/bin/bash
# build tools/panel_bench (plain + stamped) and dump the f32 chain kernel's ISA to /tmp/chain_f32.s
cd /root/repo/tools || exit 1
EXTRA="$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $EXTRA -I../bayesianlinearregressors.jl_amd/csrc panel_bench.hip -o panel_bench -save-temps=obj 2>&1 | grep -E "rror" -A3 | head -20
S=panel_bench-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN3blr18panel_chain_kernelIfLi[0-9]+EEEvPT_liiPiPjj:/{p=1} p{print} /\.Lfunc_end.*panel_chain_kernelIf/{if(p)exit}' $S > /tmp/chain_f32.s
echo "f32 kernel: $(wc -l < /tmp/chain_f32.s) lines; first barrier at line $(grep -n s_barrier /tmp/chain_f32.s | head -1 | cut -d: -f1)"
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size)" $S | tail -8 | tr '\n' ' '; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBLR_STAMPS $EXTRA -I../bayesianlinearregressors.jl_amd/csrc panel_bench.hip -o panel_bench_st 2>&1 | grep -E "rror" -A3 | head -20
rm -f panel_bench-hip-* panel_bench-host-* panel_bench.hip-hip-*
