timeout 600 python tools/stamps4.py 256 2>&1 | tail -48
timeout 600 python tools/stamps4.py 512 2>&1 | tail -3
