timeout 900 python tools/wsgemm_stress.py 2>&1 | tail -60
