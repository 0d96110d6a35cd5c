#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -k "production_randomness or untracked or graph_replay or full_size or random_configurations or saturation" 2>&1 | tail -3
timeout 300 python tools/tiles_ab.py c5 2 1 2>&1 | tail -2
python tools/ab_knobs.py --envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range --rounds 3 --draws 1 "" 2>&1 | tail -1
python tools/ab_knobs.py --envs 1024 --agents 8 --grid 512 --rounds 2 --draws 1 "" 2>&1 | tail -1
python tools/ab_knobs.py --rounds 4 --draws 8 "" 2>&1 | tail -1
