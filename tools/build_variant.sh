#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : diagnostic build of the library into objcavit_amd/lib/variants/NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p objcavit_amd/lib/variants
cp objcavit_amd/lib/libobjcavit_hip.so /tmp/ocv_keep.so 2>/dev/null || true
OCV_EXTRA_HIPCC_FLAGS="$2" python -m objcavit_amd.build --force > /dev/null
mv objcavit_amd/lib/libobjcavit_hip.so objcavit_amd/lib/variants/$1.so
[ -f /tmp/ocv_keep.so ] && mv /tmp/ocv_keep.so objcavit_amd/lib/libobjcavit_hip.so
echo built objcavit_amd/lib/variants/$1.so
