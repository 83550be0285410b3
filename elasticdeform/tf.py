"""``elasticdeform.tf`` -- the reference's TensorFlow wrapper (/root/reference/elasticdeform/tf.py) under its import
name: a re-export of :mod:`elasticdeform_amd.tf` (needs TensorFlow, like the reference's module)."""
from elasticdeform_amd.tf import deform_grid  # noqa: F401
