"""Minimal stand-in for the parts of diffusers==0.24.0 that MMGT's src/models import.

Test infrastructure only (golden-vector generation in the build container).  It lets the
reference's own `src.models.*` be imported unmodified on CPU; it is never imported by the
product (`mmgt_amd`) and never travels to the GPU box as a dependency of any test.
Semantics restated from the published diffusers 0.24.0 behaviour.
"""
from .configuration_utils import ConfigMixin, register_to_config  # noqa: F401
from .models.modeling_utils import ModelMixin  # noqa: F401
