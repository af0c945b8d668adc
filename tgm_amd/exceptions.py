"""Exception types raised at the drop-in boundary.

Same names and meaning as the reference's (tgm/exceptions.py:1-47) so callers'
``except`` clauses keep working unchanged.
"""


class TGMError(Exception):
    """Root of every error raised by this package."""


class BadHookProtocolError(TGMError):
    """The object handed to HookManager is not a DGHook."""


class BadEncoderProtocolError(TGMError):
    """The module handed to HookManager.validate_requirement is not an EncoderModule."""


class UnresolvableHookDependenciesError(TGMError):
    """requires/produces of the registered hooks admit no execution order."""


class InvalidNodeIDError(TGMError):
    """A node id collides with PADDED_NODE_ID or overflows int32."""


class EmptyGraphError(TGMError):
    """A graph without edge events was requested."""


class EventOrderedConversionError(TGMError):
    """A time-unit operation was requested on an event-ordered ('r') graph."""


class InvalidDiscretizationError(TGMError):
    """Iteration / discretization to a finer unit than the graph's own."""


class EmptyBatchError(TGMError):
    """An empty batch was produced under on_empty='raise'."""


class NativeLibraryError(TGMError):
    """libtgm_amd.so (the HIP kernels) is missing or failed; there is no CPU fallback."""
