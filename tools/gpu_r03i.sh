#!/bin/bash
# round 3, visit i: tracking bias at both pixel-centre conventions (64x1024), sequence with the reference's rules
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python tools/slam_demo.py 64 1024 6 60 2>&1 | tail -7
python tools/slam_demo.py --half-pixel 64 1024 6 60 2>&1 | tail -7
python tools/slam_demo.py sequence 64 1024 25 2>&1 | tail -12
python tools/slam_demo.py --half-pixel sequence 64 1024 25 2>&1 | tail -5
