"""l3c-pytorch_amd -- MI355X-native (gfx950 / CDNA4) L3C inference path.

Drop-in for the hot path of fab-jul/L3C-PyTorch: `MultiscaleBlueprint` / `MultiscaleNetwork.forward` / `get_P`,
`DiscretizedMixLogisticLoss`, `torchac.*`, `Bitcoding.encode/decode`, `l3c.py enc|dec` -- all running on hand-written
HIP kernels behind the C ABI declared in `include/l3c_hip.h` (`csrc/` -> `libl3c_hip.so`).  There is NO CPU fallback:
importing the compute modules without the built library, or calling them without a GPU, raises.

Layout
  csrc/          HIP kernels + extern "C" entry points (the product)
  _lib.py        ctypes binding of the C ABI (raw device pointers + hipStream_t)
  torchac.py     reference `torchac` facade (encode/decode_cdf, encode/decode_logistic_mixture)
  modules/       MultiscaleNetwork, Out, EncOut/DecOut, quantizer helpers
  criterion/     DiscretizedMixLogisticLoss, CDFOut
  bitcoding/     Bitcoding (.l3c container), coders, part-suffix helper
  blueprints/    MultiscaleBlueprint
  helpers/       .cf config parser, pad, synthetic checkpoints, paths / log-dir parsing
  auto_crop.py   recursive 2x2 tiling of large images
"""
__version__ = '0.1.0'

# Before the first HIP call: give the runtime the hardware queues `Bitcoding.encode_many`'s side-by-side forward streams need, unless
# the caller has chosen a value (helpers/runtime.py; the runtime reads the variable once, at start-up).
from .helpers import runtime as _runtime  # noqa: E402

HIP_QUEUES_CONFIGURED = _runtime.configure_hip_queues()
