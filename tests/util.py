"""Shared helpers for the parity tests: the seeded inputs live in the package (hfa_gp_amd.synthetic) so that bench.py and
__graft_entry__.smoke() do not depend on the test package; re-exported here for the tests."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hfa_gp_amd.synthetic import (FLIP_COLUMNS, INTRINSICS, look_at_label, make_inputs, perturb_state,  # noqa: E402,F401
                                  state_cpu)
