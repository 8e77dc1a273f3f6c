"""pinsage_hip -- host-side plumbing for libpinsage_hip.so (the gfx950 kernels behind the
reference's utils.random_walk / model.pinsage / model.aggregators / utils.nearest_neighbors
surface).  PyTorch is used for device memory, streams and torch.distributed only."""
from .native import lib, have_lib, LibraryMissing  # noqa: F401
