#!/bin/bash
cd /root/repo; O=gpurun_out/r05m; mkdir -p $O
timeout 300 python tools/time_filter_f64.py 2>/dev/null > $O/time_filter_f64.txt; cat $O/time_filter_f64.txt
