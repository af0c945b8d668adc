"""Sentinels shared with the reference (tgm/constants.py:3)."""
from typing import Final

PADDED_NODE_ID: Final[int] = -1
"""Node id marking an empty neighbor slot; never a valid node id."""
