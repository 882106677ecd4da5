#!/bin/bash
# A/B build of the column kernel: tools/build_variant.sh <name> [extra hipcc flags...]
#   -> kid_amd/libkidmp_<name>.so = the current thompson_column.hip (p64, nz = 120 instantiations only: KIDMP_EXPERIMENT)
#      linked with the other objects of the last full build.  For tools/ab_kernel.py; never shipped.
set -e
name=$1; shift
cd "$(dirname "$0")/../kid_amd/csrc"
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -fapprox-func -freciprocal-math -Wno-unused-const-variable -Wno-unused-function -Wno-pass-failed -mllvm -disable-machine-licm"
hipcc $FL -DKIDMP_EXPERIMENT "$@" -c thompson_column.hip -o /tmp/thompson_column_$name.o
hipcc -shared -fPIC --offload-arch=gfx950 -o ../libkidmp_$name.so thompson_host_init.o thompson_tables.o /tmp/thompson_column_$name.o \
      thompson_column_p32n.o thompson_column_f32.o table_cache.o kidmp_capi.o
ls -la ../libkidmp_$name.so
