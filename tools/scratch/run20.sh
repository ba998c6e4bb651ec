#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 tools/scratch/probe_bubble.py 2>&1 | grep -E "kind|\"value\"" | cut -c1-260
python3 tools/scratch/probe_bubble.py 2>&1 | grep -E "kind|\"value\"" | cut -c1-260
