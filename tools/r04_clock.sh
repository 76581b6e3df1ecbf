#!/bin/bash
cd $GRAFT_REPO_ROOT
B=sequential-line-search_amd/bin
SLS_WAVE_TRACE=1 $B/sequential_line_search_nd 32 30 1 2>&1 | grep "wave trace" | tail -3
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
