for v in "" NOW NOA NOGATE NOMFMA; do
  echo "=== variant ${v:-base}"
  if [ -n "$v" ]; then export OCV_LIB_PATH=$PWD/objcavit_amd/lib/variants/$v.so; fi
  OCV_PW_CFG=0,0,0 python3 tools/run_pw.py 3 4 6 8 9 10 11 12 13 2>&1 | grep "M="
done
