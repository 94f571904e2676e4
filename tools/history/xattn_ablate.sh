#!/bin/bash
# Ablation builds of the split3 cross-attention tile (tools/diag/xattn_ablation.patch.txt, one -DXA_NO_* each, built into
# objcavit_amd/lib/variants/xa_<flag>.so by hand: hipcc -D<flag> on the patched token_split3.hip + the product objects), timed at
# one batch with HIP events.  The differences against BASE localise the cost inside a tile (profiles/r03_cross_attention_roofline.txt).
# Usage: tools/xattn_ablate.sh B S [NSUB]
cd $GRAFT_REPO_ROOT
[ -n "$3" ] && export OCV_XATTN_NSUB=$3
for v in ${XA_VARIANTS:-BASE XA_NO_WLOAD XA_NO_XLOAD XA_NO_STAGE XA_NO_QMFMA XA_NO_SCORES XA_NO_CTX XA_NO_OMFMA XA_NO_STORE XA_STORE_LOCAL}; do
  OCV_LIB_PATH=objcavit_amd/lib/variants/xa_$v.so python3 tools/xattn_time.py $1 $2 2>&1 | tail -n 1 || exit 1
done
