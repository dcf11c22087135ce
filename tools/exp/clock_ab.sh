#!/bin/bash
cd /root/repo
for i in 1 2 3 4 5; do timeout -k 10 120 python3 tools/exp/clock_probe.py 2>&1 | grep "^b ="; echo; done
