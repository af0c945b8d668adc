"""Exception types raised at the drop-in boundary.

The names and meanings are the reference's (tgm/exceptions.py:1-47), so callers' ``except`` clauses keep working
unchanged; ``NativeLibraryError`` is ours.  The classes are built from one table (name, what it means) instead of
nine two-line class statements.
"""
from typing import Dict, Type


class TGMError(Exception):
    """Root of every error raised by this package."""


_MEANINGS = (
    ('BadHookProtocolError', 'The object handed to HookManager is not a DGHook.'),
    ('BadEncoderProtocolError', 'The module handed to HookManager.validate_requirement is not an EncoderModule.'),
    ('UnresolvableHookDependenciesError', 'requires/produces of the registered hooks admit no execution order.'),
    ('InvalidNodeIDError', 'A node id collides with PADDED_NODE_ID or overflows int32.'),
    ('EmptyGraphError', 'A graph without edge events was requested.'),
    ('EventOrderedConversionError', "A time-unit operation was requested on an event-ordered ('r') graph."),
    ('InvalidDiscretizationError', "Iteration / discretization to a finer unit than the graph's own."),
    ('EmptyBatchError', "An empty batch was produced under on_empty='raise'."),
    ('NativeLibraryError', 'libtgm_amd.so (the HIP kernels) is missing or failed; there is no CPU fallback.'),
)

_classes: Dict[str, Type[TGMError]] = {name: type(name, (TGMError,), {'__doc__': doc, '__module__': __name__}) for name, doc in _MEANINGS}
globals().update(_classes)
__all__ = ['TGMError', *_classes]
