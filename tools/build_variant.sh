#!/bin/bash
# tools/build_variant.sh <tag> [-DNGD_...]... -- an A/B build of the engine beside the product build:
# ngsdist_amd/libngsdist_amd.so.<tag> (git-ignored; travels to the GPU box), objects under csrc/build_<tag>/.
# A tool picks it up with NGSDIST_AMD_LIB=ngsdist_amd/libngsdist_amd.so.<tag> (ngsdist_amd/_lib.py).
set -eu
TAG=$1; shift
cd "$(dirname "$0")/../ngsdist_amd/csrc"
make -s -j8 BUILD=build_$TAG OUT=../libngsdist_amd.so.$TAG EXTRA="$*"
echo "built ngsdist_amd/libngsdist_amd.so.$TAG ($*)"
