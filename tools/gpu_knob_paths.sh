#!/bin/bash
timeout 300 python -m pytest tests -m gpu -q -x -k "untracked" 2>&1 | tail -2
IPPM_K3_DENSE=0 timeout 300 python -m pytest tests -m gpu -q -x -k "untracked or benched or golden_episode" 2>&1 | tail -2
IPPM_K3_CLASSIC=1 IPPM_NO_TILES=1 timeout 300 python -m pytest tests -m gpu -q -x -k "untracked or benched or golden_episode or tracked_area" 2>&1 | tail -2
