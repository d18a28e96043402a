#!/bin/bash
# the config 3 / 4 tests (reduced and full size), then the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -k "config3 or config4" > $O/tests_cfg.log 2>&1; echo "cfg tests rc=$?"; tail -n 5 $O/tests_cfg.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not config3 and not config4" > $O/tests_rest.log 2>&1; echo "rest rc=$?"; tail -n 3 $O/tests_rest.log
