timeout 900 python tools/race_hunt.py 2>&1 | tail -12
