for cfg in 4,1,1 2,1,1 1,1,1 4,1,2 stream; do OCV_PW_CFG=$cfg python3 tools/run_pw.py 0 1 2 3 4 5 14 2>&1 | grep "M="; done
