"""Protocol of a prior data loader (reference priors/prior.py).

A PriorDataLoader is constructed as `DataLoader(num_steps, batch_size=..., seq_len=..., **kw)` and
iterated `num_steps` times per epoch, each item being `((x[T,B,F], y[T,B]), target_y[T,B])` (or the
fused form).  Class or instance attributes: `num_features`, `num_outputs`, `fuse_x_y`; optional
`validate(model)`.  It subclasses torch's DataLoader only for isinstance() parity with the
reference -- its __init__ is never run and no workers / prefetching exist.
"""
from torch.utils.data import DataLoader


class PriorDataLoader(DataLoader):
    # protocol attributes; a prior module sets them on its loader class (`DataLoader.num_outputs = 1`) or the loader
    # instance takes them from its constructor keywords (priors/utils.py)
    num_features = None
    num_outputs = None
    fuse_x_y = False
