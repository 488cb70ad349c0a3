timeout 600 python -m pytest tests/test_stereo_gpu.py -x -q -k "data_sets or batch_group" 2>&1 | grep -v '^REBVO' | tail -30 | cut -c1-800
