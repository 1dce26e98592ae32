"""MI355X-native acoustic-model policy-gradient training path (drop-in for the hot path of
ana-kuznetsova/Policy-Gradient-ASR).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

import os as _os

# Streams of the train step (main, side, lattice, + RCCL's) should not share hardware queues; HIP's default is 4.
# Only effective if the HIP runtime has not started yet; set it yourself before importing torch otherwise.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# RCCL's peer mappings (and any device tensor shared across processes) need dmabuf IPC on this driver stack; the legacy
# mode fails with "hipIpcGetMemHandle: invalid argument".  Same caveat: read when the HSA runtime starts.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
