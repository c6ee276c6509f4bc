"""styler_amd -- MI355X-native (gfx950) STYLER forward/backward hot path.

Host side: Python mirror of the reference's `styler.STYLER` / `loss.STYLERLoss` API.
Device side: libstyler_hip.so (hand-written HIP, C ABI in include/styler_hip.h), bound with ctypes.
There is no CPU fallback: importing without the built library raises ImportError."""
from . import hparams  # noqa: F401
from ._lib import ABI_VERSION, LIB_PATH  # noqa: F401
from .runtime import rt  # noqa: F401
from .styler import STYLER  # noqa: F401
