#!/bin/bash
# Build the C restatement of the oracle (test infrastructure): oracle/c/libd2d_oracle_c.so
set -e
cd "$(dirname "$0")"
gcc -O2 -fopenmp -shared -fPIC -Wall d2d_oracle.c -o libd2d_oracle_c.so -lm
