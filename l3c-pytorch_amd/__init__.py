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

# `configure_hip_queues()` -- call it BEFORE the first HIP call in applications that code sets of differently sized images
# (`Bitcoding.encode_many`, `helpers/dataset_codec.encode_set`; `l3c.py`, `test.py` and `bench.py --config dataset` do): it asks the HIP
# runtime for the 8 hardware queues the side-by-side forward streams need (GPU_MAX_HW_QUEUES, read once at HIP start-up; helpers/runtime.py).
# NOT done on import: batches of equally sized images (the headline path) run 1.2 % FASTER with the runtime's default of four queues
# [measured, round 5: 274.7 vs 277.9 ms per step] -- without the call `encode_many` warns once and uses one forward stream.
from .helpers.runtime import configure_hip_queues  # noqa: E402,F401
