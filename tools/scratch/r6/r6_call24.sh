#!/bin/bash
# r6_call24 -- the GPU property test with 75x the examples (new window / hybrid / long-id paths under random options)
export PYTHONPATH=$PWD
DASP_HYP_EXAMPLES=6000 timeout 2400 python3 -m pytest tests/test_property.py -m gpu -x -q 2>&1 | tail -5
