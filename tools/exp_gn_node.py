#!/usr/bin/env python
"""GPU box: per-node cost of the GroupNorm shapes of a batch-8 evaluation in a replayed single-chain graph (MKD_* switches of the
library apply).    python tools/exp_gn_node.py"""
import os
import sys
sys.argv = ['x']
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'exp_floor_graph.py')).read()
exec(src[:src.index("ln(4, 64); ln(256, 1280)")])
gn(8, 1024, 320); gn(4, 1024, 320); gn(4, 1024, 960); gn(4, 1024, 640); gn(8, 256, 640); gn(4, 256, 640); gn(4, 256, 1920); gn(8, 64, 1280)
