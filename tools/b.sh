#!/bin/bash
# dev helper: build a tools/*.hip harness against the csrc headers:  tools/b.sh <name> [out-suffix] [-Dflags...]
cd /root/repo/tools || exit 1
name=$1; suf=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -I/root/repo/bayesianlinearregressors.jl_amd/csrc "$@" $name.hip -o $name$suf 2>&1 | grep -E "error" -A8 | head -40
ls -la /root/repo/tools/$name$suf | awk '{print $6,$7,$8,$9}'
