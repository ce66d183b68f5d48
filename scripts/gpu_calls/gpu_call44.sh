#!/bin/bash
python scripts/probe_gap.py 2>&1 | tail -8
