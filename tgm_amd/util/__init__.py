"""Script-level helpers the reference's examples import next to the hot path (``tgm.util``); only what a drop-in needs."""
from .seed import seed_everything

__all__ = ['seed_everything']
