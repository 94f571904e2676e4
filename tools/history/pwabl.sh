for v in "" NORES NOSTORE; do
  echo "=== variant ${v:-base}"
  if [ -n "$v" ]; then export OCV_LIB_PATH=$PWD/objcavit_amd/lib/variants/$v.so; fi
  python3 tools/run_pw.py 0 2 3 4 5 12 13 2>&1 | grep "M=" | sed 's/fp32.*| split/split/'
done
