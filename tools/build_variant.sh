#!/bin/bash
# Developer tool: a variant build of the library for before / after timing on one box (tools/ab_compare.py, CTPVAE_VARIANT_LIB):
#   bash tools/build_variant.sh <tag> [-DCTPVAE_TUNE_... ...]   ->  tools/libctpvae_radon_<tag>.bin
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
TMP=$(mktemp -d)
cp "$ROOT"/ct_pvae_amd/csrc/*.hip "$ROOT"/ct_pvae_amd/csrc/*.h "$ROOT"/ct_pvae_amd/csrc/Makefile "$TMP"/
mkdir -p "$TMP/../../include" 2>/dev/null || true
sed -i "s#../../include/ctpvae_radon.h#$ROOT/include/ctpvae_radon.h#" "$TMP"/Makefile "$TMP"/common.h
make -C "$TMP" -j8 TARGET="$ROOT/tools/libctpvae_radon_$TAG.bin" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $*" > /dev/null
rm -rf "$TMP"
ls -la "$ROOT/tools/libctpvae_radon_$TAG.bin"
