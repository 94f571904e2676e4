#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [PATCH ...] : diagnostic build of the library into objcavit_amd/lib/variants/NAME.so
# (patches default to every tools/diag/*.patch)
#
# The product sources under objcavit_amd/csrc carry no diagnostic code.  The ablation switches (-DOCV_ABL_NOLOAD,
# -DOCV_ABL_KXSHARE, -DOCV_ABL_KXRAND, -DOCV_ABL_SMALLFOOT, -DOCV_ABL_NOWRITE, -DPW_ABL_NOW / NOA / NOGATE / NOMFMA /
# NOSTORE) and the in-kernel cycle stamps (-DOCV_STAMPS) live in tools/diag/*.patch; this script applies them to a
# scratch copy of csrc/ and builds THAT, so a variant can never leak into the product library.
set -e
cd "$(dirname "$0")/.."
mkdir -p objcavit_amd/lib/variants
scratch=$(mktemp -d /tmp/ocv_csrc.XXXXXX)
trap 'rm -rf "$scratch"' EXIT
mkdir -p "$scratch/objcavit_amd/csrc" "$scratch/include"
cp objcavit_amd/csrc/* "$scratch/objcavit_amd/csrc/"
cp include/objcavit_hip.h "$scratch/include/"
name=$1; flags=$2; shift 2 || true
if [ $# -eq 0 ]; then set -- tools/diag/*.patch; fi
for p in "$@"; do patch -s -d "$scratch/objcavit_amd/csrc" -p1 < "$p"; done
# OCV_LIB_OUT: build.py writes the variant (and its objects) there; the product library is never moved or deleted
OCV_LIB_OUT="$PWD/objcavit_amd/lib/variants/$name.so" OCV_CSRC_DIR="$scratch/objcavit_amd/csrc" OCV_EXTRA_HIPCC_FLAGS="$flags" \
  python -m objcavit_amd.build --force > /dev/null
echo built objcavit_amd/lib/variants/$name.so
