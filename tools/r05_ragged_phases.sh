cd /root/repo
RAGGED=1 LAYERS=8 python tools/time_phases_y.py 128 128 4096 2>/dev/null | grep -v "^$"
