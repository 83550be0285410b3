"""``elasticdeform.torch`` (/root/reference/elasticdeform/torch.py:33): alias of
:mod:`elasticdeform_amd.torch` -- same ``deform_grid(X, displacement, *args, **kwargs)``."""
from elasticdeform_amd.torch import *  # noqa: F401,F403
from elasticdeform_amd.torch import (ElasticDeform, deform_grid, deform_random_grid,  # noqa: F401
                                     random_displacement, deform_grid_batch,
                                     deform_random_grid_batch)
