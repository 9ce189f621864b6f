#!/bin/bash
# one GPU trip: the full -m gpu suite + the default bench (scripts/gpu_suite.sh)
bash "$(dirname "$0")/gpu_suite.sh" "${1:-trip}"
