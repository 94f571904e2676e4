python3 tools/run_pw.py 2>&1 | grep "M="
echo "--- stream forced on small-N shapes"
OCV_PW_CFG=stream python3 tools/run_pw.py 1 3 4 6 14 2>&1 | grep "M="
for cfg in 4,1,1 4,2,1 4,1,2 4,2,2 2,4,1 2,2,2; do OCV_PW_CFG=$cfg python3 tools/run_pw.py 8 9 10 11 12 13 6 7 2>&1 | grep "M="; done
