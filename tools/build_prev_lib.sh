#!/bin/bash
# Builds the library of a git revision (default HEAD) as adamml_amd/libadamml_hip_prev.so: the "previous" side of tools/gpu_ab_lib.sh
# (git-ignored like every .so; travels to the GPU box).  Usage (build container): bash tools/build_prev_lib.sh [rev]
rev=${1:-HEAD}; root=$(cd "$(dirname "$0")/.." && pwd); tmp=$(mktemp -d)
git -C "$root" archive "$rev" adamml_amd/csrc include | tar -x -C "$tmp" || exit 1
cd "$tmp" || exit 1
for f in adamml_amd/csrc/*.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -w -c "$f" -o "${f%.hip}.o" & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/adamml_amd/libadamml_hip_prev.so" adamml_amd/csrc/*.o && echo "built libadamml_hip_prev.so from $rev"
rm -rf "$tmp"
