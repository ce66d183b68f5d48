#!/bin/bash
for d in 2 0; do echo "== PLNLP_STEP_THROTTLE=$d"; PLNLP_STEP_THROTTLE=$d python scripts/probe_gap.py 2>&1 | grep -E "ms/step|join"; done
