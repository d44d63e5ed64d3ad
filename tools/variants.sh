#!/bin/bash
# Builds measurement variants of libflatgfa.so side by side (in this container; the .so files travel to the GPU box):
#   tools/variants.sh tag1 "-DFOO=1" tag2 "-DBAR=2" ...   ->  pollen_amd/lib_<tag>/libflatgfa.so
# Select one with FLATGFA_LIB=pollen_amd/lib_<tag>/libflatgfa.so.
cd "$(dirname "$0")/../pollen_amd/csrc"
while [ $# -ge 2 ]; do
  tag=$1; extra=$2; shift 2
  make -j8 LIBDIR=../lib_$tag OBJDIR=../build_$tag BINDIR=../bin_$tag EXTRA="$extra" ../lib_$tag/libflatgfa.so ../build_$tag/pinned.ok 2>&1 | grep -E "error|warning: unused|Error|FAILED" 
  ls -la ../lib_$tag/libflatgfa.so
done
