#!/bin/bash
# build a variant of the library next to the product one (never replaces it): tools/r6_buildlib.sh <name> [hipcc flags...]
#   -> build/ab/<name>.so, used through UNIRES_LIB for same-box A/B runs and instrumented (profiling / ablation) builds
set -e
cd "$(dirname "$0")/.."
name=$1; shift
obj=/tmp/objs_$name; mkdir -p $obj build/ab
srcs="ops fused splat splat2 ata1 pull2 aligned shift stencil orient cg admm fftpre api"
# (per-source flags as in __graft_entry__.py; NO_SLP="" builds every source with the vectoriser on)
noslp=${NO_SLP-"splat2 ata1 stencil shift pull2"}
for s in $srcs; do
  extra=""; for n in $noslp; do [ "$n" = "$s" ] && extra="-fno-slp-vectorize"; done
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function $extra "$@" -c unires_amd/csrc/$s.hip -o $obj/$s.o 2>/dev/null || echo "FAILED $s" ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib $(for s in $srcs; do echo $obj/$s.o; done) -o build/ab/$name.so
ls -la build/ab/$name.so
