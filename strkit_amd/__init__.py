"""strkit_amd — MI355X (gfx950) backend for the per-read repeat-count hot path of ``strkit call``.

Scope: strkit/call/repeats.py::get_repeat_count and the per-read loop around it
(strkit/call/call_locus.py:1082-1161) — nothing else of STRkit.  See DESIGN.md.
"""
from .repeat_count_params import RepeatCountParams, default_read_rc_params, get_reference_rc_params  # noqa: F401

__version__ = "0.1.0"
