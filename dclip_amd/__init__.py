"""dclip_amd — MI355X-native DCLIP distillation step."""
import os as _os

# The pool's host driver only supports dmabuf IPC; RCCL / device-tensor sharing across ranks fails with
# `hipIpcGetMemHandle: invalid argument` otherwise.  It has to be in the environment before the process's first HIP
# call, so it is defaulted at package import (importing torch does not initialise HIP).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.2.0"
