"""posepaf -- MI355X-native post-processing for the SimplePose / Improved-Body-Parts bottom-up pose pipeline.

Layout: csrc/ (HIP kernels + C ABI, built into posepaf/libposepaf.so), posepaf/ (this package: ctypes binding,
device pipeline, skeleton constants, synthetic scenes), and next to it the reference-shaped modules
(utils/pafprocess, utils/parse_skeletons, ...) that keep the reference's call surface."""
from . import skeleton  # noqa: F401

__all__ = ["skeleton"]
