"""Search parameters of the repeat counter — same names, defaults and schedule as the reference
(strkit/call/repeat_count_params.py:9-42; defaults strkit/call/params.py:26-27,43-45,157-163)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

__all__ = ["RepeatCountMethod", "RepeatCountParams", "get_reference_rc_params", "default_read_rc_params"]

RepeatCountMethod = Literal["repalign", "comp"]


@dataclass(frozen=True)
class RepeatCountParams:
    method: RepeatCountMethod
    max_iters: int
    initial_local_search_range: int
    initial_step_size: int


def default_read_rc_params() -> RepeatCountParams:
    """CallParams.rc_params with the CLI defaults (params.py:157-163): repalign, 50 iters, range 3, step 1."""
    return RepeatCountParams(method="repalign", max_iters=50, initial_local_search_range=3, initial_step_size=1)


def get_reference_rc_params(method: RepeatCountMethod, ref_est_cn: int, default_ref_max_iters: int) -> RepeatCountParams:
    """Reference-side schedule: search less, in bigger steps, for large estimated copy numbers
    (repeat_count_params.py:17-42)."""
    max_iters, step, lsr = default_ref_max_iters, 1, 3
    if ref_est_cn >= 2000:
        max_iters, step, lsr = 50, 15, 1
    elif ref_est_cn >= 1000:
        max_iters, step = 150, 5
    elif ref_est_cn >= 200:
        max_iters, step = 200, 3
    return RepeatCountParams(method=method, max_iters=max_iters, initial_local_search_range=lsr,
                             initial_step_size=step)
