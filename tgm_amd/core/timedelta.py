"""Time-granularity algebra used by the loader (tgm/core/timedelta.py:10-112).

'r' is the event-ordered pseudo-unit (no wall-clock meaning); every other unit
is a fixed number of nanoseconds and carries an integer multiplier.
"""
from __future__ import annotations

from dataclasses import dataclass

from ..exceptions import EventOrderedConversionError

_NANOS = {
    'ns': 1,
    'us': 10**3,
    'ms': 10**6,
    's': 10**9,
    'm': 60 * 10**9,
    'h': 3600 * 10**9,
    'D': 86400 * 10**9,
    'W': 7 * 86400 * 10**9,
    'M': 30 * 86400 * 10**9,
    'Y': 365 * 86400 * 10**9,
}


@dataclass(frozen=True)
class TimeDeltaDG:
    unit: str
    value: int = 1

    def __post_init__(self) -> None:
        if not isinstance(self.value, int) or isinstance(self.value, bool) or self.value <= 0:
            raise ValueError(f'Value must be a positive integer, got: {self.value}')
        if self.unit == 'r':
            if self.value != 1:
                raise ValueError('Only value=1 is supported for event-ordered TimeDeltaDG')
        elif self.unit not in _NANOS:
            raise ValueError(f"Unknown unit: {self.unit}, expected one of {['r'] + list(_NANOS)}")

    @property
    def is_event_ordered(self) -> bool:
        return self.unit == 'r'

    @property
    def is_time_ordered(self) -> bool:
        return self.unit != 'r'

    def convert(self, other: 'str | TimeDeltaDG') -> float:
        """How many ``other`` ticks one tick of ``self`` spans."""
        if isinstance(other, str):
            other = TimeDeltaDG(other)
        if self.is_event_ordered or other.is_event_ordered:
            raise EventOrderedConversionError('Cannot compare granularity for event-ordered TimeDeltaDG')
        # The reference's two roundings, in its order (tgm/core/timedelta.py:99-112): the VALUE ratio in float first, then times / over
        # the exact integer ratio of the units -- not the correctly rounded quotient, which differs from it in the last place for
        # e.g. 3 s -> 5 D (found by the reference's own test_timedelta.py) and would move `discretize`'s bucket edges with it.
        mine, theirs = _NANOS[self.unit], _NANOS[other.unit]
        value_ratio = self.value / other.value
        return value_ratio * (mine // theirs) if mine > theirs else value_ratio / (theirs // mine)

    def is_coarser_than(self, other: 'str | TimeDeltaDG') -> bool:
        return self.convert(other) > 1


# The reference keeps the native unit of every TGB / TGB-Seq data set next to this class (tgm/core/timedelta.py:115-149) for its ``from_tgb``
# loaders.  Ingest from TGB is out of scope here, so the tables are empty: the names exist because code written
# against the reference imports them from this module.
TGB_TIME_DELTAS: dict = {}
TGB_SEQ_TIME_DELTAS: dict = {}
