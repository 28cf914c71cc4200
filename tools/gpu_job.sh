#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/prefix_debug.py 2>&1 | tail -8
