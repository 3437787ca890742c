"""The schedule-type string grammar of the reference scheduler
(vllm/core/scheduler.py:268-331), e.g. ``opt-125m-sharegpt-starv200-period10``.

Only the predictor-ordered policies are in scope here; the parse of
``starv<S>-period<P>`` uses the reference's own slicing arithmetic so odd strings
behave identically (including its failure modes: ``starv`` without ``period`` raises
``ValueError`` there too).
"""
from __future__ import annotations

import dataclasses

# prefix -> (ordering name, needs predictor score)      scheduler.py:290-331
_POLICIES = (("fifo", False), ("srtf", False), ("FAKEPO", False), ("PO", False),
             ("xpt", True), ("tpt", True), ("opt", True))


@dataclasses.dataclass(frozen=True)
class ScheduleType:
    raw: str
    policy: str          # "opt", "tpt", ...
    need_score: bool
    starv: int           # -1 = starvation control off (scheduler.py:270)
    period: int
    table_path: str = ""  # xpt: the path between { and } of the string, torch.load'ed at scheduler.py:312

    @property
    def uses_priority_key(self) -> bool:      # scheduler.py:996 vs :998
        return self.starv != -1


def parse_schedule_type(schedule_type: str) -> ScheduleType:
    starv, period = -1, 0
    if "starv" in schedule_type:                                   # scheduler.py:271-275
        starv = int(schedule_type[schedule_type.find("starv") + len("starv"):
                                  schedule_type.find("period") - 1])
        period = int(schedule_type[schedule_type.find("period") + len("period"):])
    for prefix, need in _POLICIES:
        if schedule_type.startswith(prefix):
            path = ""
            if prefix == "xpt" and "{" in schedule_type and "}" in schedule_type:
                # scheduler.py:312 (same slicing: find / rfind).  Without braces the reference's slice is garbage
                # and its torch.load fails; here the path stays empty and the ranker asks for the table instead.
                path = schedule_type[schedule_type.find("{") + 1:schedule_type.rfind("}")]
            return ScheduleType(schedule_type, prefix, need, starv, period, path)
    if schedule_type.startswith("fcfs") or schedule_type in ("sjf", "ljf"):
        return ScheduleType(schedule_type, "fcfs", False, starv, period)
    raise AssertionError(f"Not Supported Schedule Type {schedule_type}")   # scheduler.py:331
