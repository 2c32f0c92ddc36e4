"""Training half of the hot path (SURVEY section 8 rows b-callers, e-train, f2): a differentiable forward whose convolutions
(98 % of the FLOPs) run on the HIP kernels in both directions, the Charbonnier losses of the reference, and a data-parallel
step with a flat-buffer gradient all-reduce over RCCL."""
from .ops import conv2d                      # noqa: F401
from .graph import forward_train             # noqa: F401
from .loss import charbonnier_loss, charbonnier_loss_mmedit   # noqa: F401
from .step import FlatGradAllReduce, TrainStep                # noqa: F401
