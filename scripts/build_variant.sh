#!/bin/bash
# build a diagnostic variant of libipdm.so: scripts/build_variant.sh <tag> <file.hip> "<extra hipcc flags>" -> _variants/libipdm_<tag>.so
# (only <file.hip> is recompiled with the flags; the production library is rebuilt afterwards)
set -e
TAG=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
mkdir -p _variants
touch inverseproblemwithdiffusionmodel_amd/csrc/$SRC
IPDM_EXTRA_HIPCC_FLAGS="$FLAGS" python -c "from inverseproblemwithdiffusionmodel_amd.csrc import build; build.build()"
cp inverseproblemwithdiffusionmodel_amd/libipdm.so _variants/libipdm_$TAG.so
touch inverseproblemwithdiffusionmodel_amd/csrc/$SRC
python -c "from inverseproblemwithdiffusionmodel_amd.csrc import build; build.build()"
