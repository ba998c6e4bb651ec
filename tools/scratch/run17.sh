#!/bin/bash
cd $GRAFT_REPO_ROOT
P=tools/scratch/probe_launch.py
echo "== default"; python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== queues 2"; DEBUG_HIP_FORCE_GRAPH_QUEUES=2 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== queues 8"; DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== packet capture 0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== packet capture 1"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== batch 16"; DEBUG_HIP_GRAPH_BATCH_SIZE=16 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== batch 256"; DEBUG_HIP_GRAPH_BATCH_SIZE=256 python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
echo "== default again"; python3 $P 2>&1 | grep -E "host replay|\"value\"" | cut -c1-200
