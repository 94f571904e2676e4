"""Tensor-level wrappers over the C ABI (include/objcavit_hip.h).

PyTorch is used here for device memory, the current HIP stream and nothing
else: every function validates its operands on the host (shape, dtype, device,
contiguity -- a wrong shape must never reach a hand-written kernel), takes raw
``data_ptr()`` values and enqueues our kernels on ``torch.cuda.current_stream()``.
All of them raise if the tensors are not on a GPU or the library is missing.

One module per kernel family (round 5; the single 2 100-line file of rounds 1 - 4):
    _core    operand checks, timing hooks, eager islands, forks / side streams, workspace store, fp16-pair control state (range guard)
    tokens   linear, LayerNorm, attention, multi-head attention, transformer encoder layers
    conv     hl32 split activations, split implicit-GEMM / Winograd / tap-form / exact convolutions, resize + concat + split
    heads    patch embedding, pixel-wise dot, bin head, bin edges, ragged object lists, positional-embedding samplers
    encoder  EfficientNet NHWC blocks (stem, 1x1, depthwise + squeeze-excite, fused expand + depthwise), validation metrics
Every name stays reachable as ``hip_ops.<name>`` (this file re-exports the five modules' namespaces; state objects such as
``ROUTE_REPORT`` / ``_TLS`` are shared, not copied).
"""
from ._core import *       # noqa: F401,F403
from .tokens import *      # noqa: F401,F403
from .conv import *        # noqa: F401,F403
from .heads import *       # noqa: F401,F403
from .encoder import *     # noqa: F401,F403
