#!/bin/bash
python scripts/probe_warm.py 2>&1 | tail -31
