"""Exception types of the drop-in surface.

Same names and message behaviour as the reference's ``probables/exceptions.py:4-92`` (only the ones the
accelerated path can raise), so ``except InitializationError`` written against pyprobables keeps working.
"""


class ProbablesBaseException(Exception):
    """root of the hierarchy; ``str(exc)`` is the message itself"""

    def __init__(self, message: str) -> None:
        super().__init__(message)
        self.message = message

    def __str__(self) -> str:
        return self.message


class InitializationError(ProbablesBaseException):
    """bad or insufficient constructor parameters"""


class NotSupportedError(ProbablesBaseException):
    """the operation is not available for this structure"""


class SimilarityError(ProbablesBaseException):
    """two filters cannot be combined (different size / hash family)"""


class RotatingBloomFilterError(ProbablesBaseException):
    """popping the last filter of a RotatingBloomFilter (reference exceptions.py:62-70)"""


class CountMinSketchError(ProbablesBaseException):
    """mismatched count-min sketches in ``join``"""


class NativeLibraryError(RuntimeError):
    """libpsk_hip.so (the HIP engine) is missing, failed to load, or reported an error.

    There is deliberately NO CPU fallback behind the data path: without the engine every
    add/check raises this."""
