"""cassie_amd -- Python access to the MI355X-native batched Cassie physics library."""
from .phys import Batch, Model, model_path  # noqa: F401
