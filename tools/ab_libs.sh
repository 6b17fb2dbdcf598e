#!/usr/bin/env bash
# A/B of two builds of the library in one GPU session, interleaved (tools/ab_old_new.py): usage ab_libs.sh libA.so libB.so
python3 tools/ab_old_new.py "$1" "$2"
