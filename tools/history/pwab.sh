for v in BASE NEW; do
  if [ "$v" = BASE ]; then export OCV_LIB_PATH=$PWD/objcavit_amd/lib/variants/BASE.so; else unset OCV_LIB_PATH; fi
  echo "== $v"; python3 tools/run_pw.py 3 4 6 7 8 9 10 11 12 13 2>&1 | grep "M=" | sed "s/fp32.*| split/split/"
done
