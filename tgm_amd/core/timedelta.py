"""Time-granularity algebra used by the loader (tgm/core/timedelta.py:10-112).

'r' is the event-ordered pseudo-unit (no wall-clock meaning); every other unit
is a fixed number of nanoseconds and carries an integer multiplier.
"""
from __future__ import annotations

from dataclasses import dataclass
from fractions import Fraction

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
        return float(Fraction(self.value * _NANOS[self.unit], other.value * _NANOS[other.unit]))

    def is_coarser_than(self, other: 'str | TimeDeltaDG') -> bool:
        return self.convert(other) > 1
